#!/usr/bin/env python3
"""Developer sanity run on a GPU box: K1 parity, early-transition agreement with the oracle, timing."""
import contextlib
import io
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import oracle  # noqa: E402
from biolith_amd.engine import OccuDataset  # noqa: E402
from biolith_amd.models import simulate  # noqa: E402


def sim(**kw):
    with contextlib.redirect_stdout(io.StringIO()):
        return simulate(**kw)


def check_logp(name, data, nb=4):
    od = oracle.OracleData(data["site_covs"], data["obs_covs"], data["obs"])
    ds = OccuDataset(data["site_covs"], data["obs_covs"], data["obs"])
    rng = np.random.default_rng(1)
    th = rng.uniform(-2, 2, size=(nb, od.D)).astype(np.float32).astype(np.float64)
    Uo, Go = od.potential_grad(th)
    for staged in (True, False):
        Ug, Gg = ds.logp_grad(th, staged=staged)
        print(f"[{name}] staged={staged} U rel err {np.max(np.abs(Ug - Uo) / np.abs(Uo)):.3e}  "
              f"grad rel err {np.max(np.abs(Gg - Go)) / np.max(np.abs(Go)):.3e}   U0={Uo[0]:.6f} / {Ug[0]:.6f}")
    return od, ds


def main():
    d_small, _ = sim(n_sites=300, n_site_covs=3, n_obs_covs=3, deployment_days_per_site=35, session_duration=7)
    od, ds = check_logp("small_3x3", d_small)
    d_def, _ = sim()
    check_logp("default", d_def)
    d_miss, _ = sim(simulate_missing=True, n_periods=3)
    check_logp("missing_3p", d_miss)

    # early transitions vs oracle (same RNG streams)
    W, S = 30, 20
    o = oracle.nuts_run(od, num_warmup=W, num_samples=S, num_chains=2, seed=5, trace=True)
    r = ds.nuts(num_warmup=W, num_samples=S, num_chains=2, seed=5)
    print("oracle steps", o["num_steps"][0][:20])
    print("gpu    steps", r.num_steps[0][:20])
    print("oracle draw0", o["draws"][0][0])
    print("gpu    draw0", r.draws[0][0])
    print("oracle eps", o["step_size"], "gpu eps", r.step_size)
    print("gpu kernel ms", r.kernel_ms, "k", r.wgs_per_chain, "lds", r.lds_bytes, r.lds_staged, "nleap", r.n_leapfrog.tolist())

    # cfg2
    d2, tp = sim(n_sites=10000, n_site_covs=3, n_obs_covs=3, deployment_days_per_site=35, session_duration=7)
    od2, ds2 = check_logp("cfg2", d2)
    for trial in range(3):
        t0 = time.time()
        r2 = ds2.nuts(num_warmup=1000, num_samples=1000, num_chains=4, seed=trial)
        wall = time.time() - t0
        nl = r2.n_leapfrog.sum()
        print(f"cfg2 trial {trial}: wall {wall*1e3:.1f} ms kernel {r2.kernel_ms:.1f} ms  leapfrogs {nl} "
              f"({r2.kernel_ms*1e3/ (nl/4):.2f} us/leapfrog/chain) k={r2.wgs_per_chain} staged={r2.lds_staged} "
              f"eps {r2.step_size} div {r2.diverging.sum()} mean steps {r2.num_steps.mean():.2f}")
        print("   mean", r2.draws.reshape(-1, 8).mean(0))
        print("   sd  ", r2.draws.reshape(-1, 8).std(0))
    print("   true", np.concatenate([tp["beta"][0], tp["alpha"][0]]))
    for k in (5, 10, 40):
        r3 = ds2.nuts(num_warmup=1000, num_samples=1000, num_chains=4, seed=0, wgs_per_chain=k)
        nl = r3.n_leapfrog.sum()
        print(f"cfg2 k={k}: kernel {r3.kernel_ms:.1f} ms ({r3.kernel_ms*1e3/(nl/4):.2f} us/leapfrog/chain) staged={r3.lds_staged}")


if __name__ == "__main__":
    main()
