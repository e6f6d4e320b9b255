#!/usr/bin/env python3
"""us per leapfrog of the shapes that run the sampler's GRP instantiation (3 + 3 covariates, so that (3, 3)-only variant libraries serve):
   python tools/time_grp_shapes.py [lib.so ...]"""
import contextlib, io, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
if len(sys.argv) > 1 and sys.argv[1] == "child":
    import numpy as np
    from biolith_amd.engine import OccuDataset
    from biolith_amd.models import simulate
    shapes = [("100x52 (2 chains)", dict(n_sites=100), 2), ("200x10", dict(n_sites=200, deployment_days_per_site=70), 4),
              ("stacked 2000x8x4", dict(n_sites=2000, n_periods=8, deployment_days_per_site=28), 4), ("5000x10", dict(n_sites=5000, deployment_days_per_site=70), 4),
              ("1600x32 (1 chain)", dict(n_sites=1600, deployment_days_per_site=7 * 32), 1), ("6400x64 (1 chain)", dict(n_sites=6400, deployment_days_per_site=7 * 64), 1)]
    for name, kw, chains in shapes:
        with contextlib.redirect_stdout(io.StringIO()):
            d, _ = simulate(n_site_covs=3, n_obs_covs=3, session_duration=7, **kw)
        ds = OccuDataset(d["site_covs"], d["obs_covs"], d["obs"])
        us = []
        for s in range(5):
            r = ds.nuts(num_warmup=500, num_samples=500, num_chains=chains, seed=s)
            us.append(1e3 * r.kernel_ms / r.n_leapfrog.sum(axis=1).max())
        print(f"  {name:20s} lanes {r.lane_group} k={r.wgs_per_chain:2d} thr={r.threads_per_wg}  {np.median(us[1:]):.3f} us/leapfrog", flush=True)
        ds.close()
else:
    for lib in sys.argv[1:] or [os.path.join("biolith_amd", "lib", "libbiolith_hip.so")]:
        print(lib, flush=True)
        subprocess.run([sys.executable, os.path.abspath(__file__), "child"], env=dict(os.environ, BIOLITH_HIP_LIB=os.path.join(ROOT, lib)))
