#!/usr/bin/env python3
"""Dynamic occupancy (builder-defined; BASELINE.json configs[4] as worded) at 2 000 sites x 8 seasons x 4 visits, 3 + 3 covariates,
4 chains: K1 parity against the float64 oracle and the time per leapfrog.   python tools/time_dyn.py [lib.so ...]
(BIOLITH_HIP_DYN_G = 1 / 2 / 4 / 8 overrides the lanes per site pair)"""
import contextlib, io, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
if len(sys.argv) > 1 and sys.argv[1] == "child":
    import numpy as np
    import oracle
    from biolith_amd.engine import OccuDataset
    from biolith_amd.models import simulate_dyn
    with contextlib.redirect_stdout(io.StringIO()):
        d, truth = simulate_dyn(n_sites=2000, n_periods=8, n_site_covs=3, n_obs_covs=3, deployment_days_per_site=28, session_duration=7)
    od = oracle.OracleData(d["site_covs"], d["obs_covs"], d["obs"], model="occu_dyn")
    ds = OccuDataset(d["site_covs"], d["obs_covs"], d["obs"], model="occu_dyn")
    th = np.random.default_rng(1).uniform(-2, 2, size=(4, od.D)).astype(np.float32).astype(np.float64)
    Uo, Go = od.potential_grad(th)
    Ug, Gg = ds.logp_grad(th)
    print("  K1 rel dU", np.max(np.abs(Ug - Uo) / np.abs(Uo)), " max|dG|/max|G|", float(np.max(np.abs(Gg - Go)) / np.max(np.abs(Go))))
    for s in range(3):
        r = ds.nuts(num_warmup=1000, num_samples=1000, num_chains=4, seed=s)
        per_chain = r.n_leapfrog.sum(axis=1)
        print(f"  cfg5 dyn seed {s}: kernel {r.kernel_ms:8.2f} ms  {1e3 * r.kernel_ms / per_chain.max():6.2f} us/leapfrog (slowest chain {int(per_chain.max())} of {int(per_chain.sum())})"
              f"  k={r.wgs_per_chain} threads {r.threads_per_wg} div {int(r.diverging.sum())} l2local {r.chains_l2_local}")
else:
    libs = sys.argv[1:] or [os.path.join("biolith_amd", "lib", "libbiolith_hip.so")]
    for lib in libs:
        print(lib, "DYN_G =", os.environ.get("BIOLITH_HIP_DYN_G", "auto"), flush=True)
        subprocess.run([sys.executable, os.path.abspath(__file__), "child"], env=dict(os.environ, BIOLITH_HIP_LIB=os.path.join(ROOT, lib)))
