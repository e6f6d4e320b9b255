#!/usr/bin/env python3
"""us per leapfrog of rows of the reference's benchmark grid (one chain, 100 + 500) at 3 + 3 covariates (what A/B variant libraries are built
for), for variant libraries:   python tools/time_grid_rows.py lib.so [lib.so ...] [-- row ...]"""
import contextlib, io, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
if len(sys.argv) > 1 and sys.argv[1] == "child":
    import numpy as np
    from biolith_amd.engine import OccuDataset
    from biolith_amd.models import simulate
    for i in [int(a) for a in sys.argv[2:]]:
        n_sites, visits = int(100 * 2 ** i), int(8 * 2 ** (i / 2))
        with contextlib.redirect_stdout(io.StringIO()):
            data, _ = simulate(n_site_covs=3, n_obs_covs=3, n_sites=n_sites, deployment_days_per_site=visits * 7, session_duration=7, simulate_missing=False, random_seed=42 + i)
        ds = OccuDataset(data["site_covs"], data["obs_covs"], data["obs"])
        us = []
        for s in range(3):
            r = ds.nuts(num_warmup=100, num_samples=500, num_chains=1, seed=s)
            us.append(1e3 * r.kernel_ms / (int(r.n_leapfrog.sum()) + 1))
        print(f"  row {i} {n_sites} x {visits}: {min(us):.3f} us/leapfrog (best of 3: {' '.join(f'{u:.3f}' for u in us)}) k={r.wgs_per_chain} lanes/pair {r.lane_group} {r.kernel_name.strip()}", flush=True)
        ds.close()
else:
    args = sys.argv[1:]; rows = ["6", "7"]
    if "--" in args:
        j = args.index("--"); rows = args[j + 1:]; args = args[:j]
    for lib in args or [os.path.join("biolith_amd", "lib", "libbiolith_hip.so")]:
        print(lib, flush=True)
        subprocess.run([sys.executable, os.path.abspath(__file__), "child"] + rows, env=dict(os.environ, BIOLITH_HIP_LIB=os.path.join(ROOT, lib)))
