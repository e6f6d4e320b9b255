#!/usr/bin/env python3
"""Lanes per site pair (occu_device.hpp: bl_eval_sites_grp) on the many-visits shapes: K1 against the oracle and the time per
leapfrog, for the forced group sizes BIOLITH_HIP_OCCU_G = 1 (one pair per lane: the kernels of rounds 1-3), 2, 4, 8, 16 and for the
host's own choice.

    python tools/time_occu_g.py [--quick]

Shapes: BASELINE configs[0] (simulate() defaults: 100 x 52, 2 chains), the stacked-period stand-in of configs[4] (2 000 x 8 x 4,
4 chains), 5 000 x 10 (4 chains), the headline (10 000 x 5, 4 chains: must stay where it is) and rows of the reference's benchmark
grid (benchmarks/occu_spoccupancy.py:16-70: one chain).
"""
import contextlib, io, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import oracle
from biolith_amd.engine import OccuDataset
from biolith_amd.models import simulate


def sim(**kw):
    with contextlib.redirect_stdout(io.StringIO()):
        d, _ = simulate(**kw)
    return d


def shapes(quick):
    s = [("cfg1 100x52 (2 chains)", sim(), 2, {}),
         ("small 200x10 3+3", sim(n_sites=200, n_site_covs=3, n_obs_covs=3, deployment_days_per_site=70, session_duration=7), 4, {}),
         ("test 100x52x3periods", sim(n_periods=3, simulate_missing=True), 2, {}),
         ("stacked 2000x8x4", sim(n_sites=2000, n_periods=8, n_site_covs=3, n_obs_covs=3, deployment_days_per_site=28, session_duration=7), 4, {}),
         ("occu 5000x10", sim(n_sites=5000, n_site_covs=3, n_obs_covs=3, deployment_days_per_site=70, session_duration=7), 4, {}),
         ("headline 10000x5", sim(n_sites=10000, n_site_covs=3, n_obs_covs=3, deployment_days_per_site=35, session_duration=7), 4, {})]
    if not quick:
        for i in (2, 4, 5, 6):  # grid rows: 100 * 2^i sites x int(8 * 2^(i/2)) visits, one covariate each, one chain
            n, j = 100 * 2 ** i, int(8 * 2 ** (i / 2))
            s.append((f"grid {n}x{j} (1 chain)", sim(n_sites=n, deployment_days_per_site=7 * j, session_duration=7, random_seed=i), 1, {}))
        s.append(("occu_fp 2000x8x4", sim(n_sites=2000, n_periods=8, n_site_covs=3, n_obs_covs=3, deployment_days_per_site=28, session_duration=7,
                                           prob_fp_constant=0.1), 4, dict(model="occu_fp", fp_mode="constant")))
    return s


def main():
    quick = "--quick" in sys.argv
    for name, d, chains, kw in shapes(quick):
        fp = kw.get("model") == "occu_fp"
        od = None if fp else oracle.OracleData(d["site_covs"], d["obs_covs"], d["obs"])
        ds = OccuDataset(d["site_covs"], d["obs_covs"], d["obs"], **kw)
        th = np.random.default_rng(1).uniform(-1.5, 1.5, size=(3, ds.D)).astype(np.float32).astype(np.float64)
        if fp:
            th[:, -1] = -2.0
        ref = od.potential_grad(th) if od is not None else None
        base_steps = None
        for g in ("1", "2", "4", "8", "16", "", "multi"):
            # "": the host's own choice (small problems: ONE workgroup of 7 compute waves, no exchange); "multi": its choice with
            # that form switched off (BIOLITH_HIP_SINGLE=0); the forced sizes run on the multi-workgroup form too
            os.environ["BIOLITH_HIP_SINGLE"] = "1" if g == "" else "0"
            if g and g != "multi":
                os.environ["BIOLITH_HIP_OCCU_G"] = g
            else:
                os.environ.pop("BIOLITH_HIP_OCCU_G", None)
            try:
                U, G = ds.logp_grad(th)
                if ref is None:
                    ref = (U, G)  # (false positives: against the one-pair-per-lane form)
                eu = np.max(np.abs(U - ref[0]) / np.abs(ref[0])); eg = np.max(np.abs(G - ref[1])) / np.max(np.abs(ref[1]))
                ds.nuts(num_warmup=300, num_samples=300, num_chains=chains, seed=0)
                r = ds.nuts(num_warmup=300, num_samples=300, num_chains=chains, seed=1)
            except Exception as e:  # noqa: BLE001
                print(f"{name:28s} G={g or 'auto':4s} FAILED: {e}")
                continue
            per_chain = r.n_leapfrog.sum(axis=1)
            steps = r.num_steps[:, :5].ravel().tolist()
            if base_steps is None:
                base_steps = steps
            print(f"{name:28s} G={g or 'auto':4s} lanes(t,j)={r.lane_group} k={r.wgs_per_chain:3d} thr={r.threads_per_wg} "
                  f"{1e3 * r.kernel_ms / per_chain.max():6.2f} us/leapfrog  kernel {r.kernel_ms:8.2f} ms  K1 dU {eu:.1e} dG {eg:.1e}  "
                  f"first trees {'same' if steps == base_steps else 'DIFFER'}", flush=True)
        ds.close()


if __name__ == "__main__":
    main()
