#!/usr/bin/env python3
"""A/B of (3,3)-only variant libraries on the headline workload (each in its own process):
   make -C biolith_amd/csrc variant NAME=x [EXTRA=-D...]   ->  biolith_amd/lib/libbiolith_hip_x.so
   python tools/ab_libs.py biolith_amd/lib/libbiolith_hip_a.so biolith_amd/lib/libbiolith_hip_b.so
Also checks the first trees against the oracle on a small problem."""
import contextlib, io, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
if len(sys.argv) > 1 and sys.argv[1] == "child":
    import numpy as np
    import oracle
    from biolith_amd.engine import OccuDataset
    from biolith_amd.models import simulate
    with contextlib.redirect_stdout(io.StringIO()):
        small, _ = simulate(n_sites=700, n_site_covs=3, n_obs_covs=3, deployment_days_per_site=35, session_duration=7, random_seed=4)
        data, _ = simulate(n_sites=10000, n_site_covs=3, n_obs_covs=3, deployment_days_per_site=35, session_duration=7)
    od = oracle.OracleData(small["site_covs"], small["obs_covs"], small["obs"])
    ds = OccuDataset(small["site_covs"], small["obs_covs"], small["obs"])
    o = oracle.nuts_run(od, 160, 40, num_chains=3, seed=5)
    r = ds.nuts(num_warmup=160, num_samples=40, num_chains=3, seed=5)
    print("  small: first trees equal", bool(np.array_equal(o["num_steps"][:, :4], r.num_steps[:, :4])), "same-steps frac", float((o["num_steps"] == r.num_steps).mean()),
          "step size", np.round(o["step_size"], 4).tolist(), np.round(r.step_size, 4).tolist(), "nleap", o["n_leapfrog"].sum(), r.n_leapfrog.sum())
    ds = OccuDataset(data["site_covs"], data["obs_covs"], data["obs"])
    ms, leaps = [], []
    for s in range(12):
        r = ds.nuts(num_warmup=1000, num_samples=1000, num_chains=4, seed=s)
        ms.append(r.kernel_ms)
        leaps.append(int(r.n_leapfrog.sum()) + 4)
    nl = leaps[-1]
    import hashlib
    r0 = ds.nuts(num_warmup=1000, num_samples=1000, num_chains=4, seed=0)
    print("  cfg2 seed 0: sha256(draws | num_steps | step_size | inv_mass) =", hashlib.sha256(r0.draws.tobytes() + r0.num_steps.tobytes() + r0.step_size.tobytes() + r0.inv_mass.tobytes()).hexdigest()[:24],
          "threads per workgroup", r0.threads_per_wg)
    print(f"  cfg2: kernel ms {np.round(ms, 2).tolist()}  ALL (after the first): {1e3 * sum(ms[1:]) / (sum(leaps[1:]) / 4):.4f} us/leapfrog/chain;  last: {1e3 * r.kernel_ms / (nl / 4):.3f} us/leapfrog/chain, div {int(r.diverging.sum())}, "
          f"l2local {r.chains_l2_local}, coef means {r.draws.reshape(-1, 8).mean(0).round(4).tolist()}")
else:
    for lib in sys.argv[1:]:
        print(lib, flush=True)
        subprocess.run([sys.executable, os.path.abspath(__file__), "child"], env=dict(os.environ, BIOLITH_HIP_LIB=os.path.join(ROOT, lib)))
