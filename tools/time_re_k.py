#!/usr/bin/env python3
"""Random effects at the bench shape (site effects, 10 000 sites x 10 visits, D = 20 009): us per leapfrog of the slowest chain over the
workgroups per chain (BIOLITH_HIP_RE_WGS; the host's rule takes 32).   python tools/time_re_k.py [k ...]"""
import contextlib, io, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from biolith_amd.engine import OccuDataset
from biolith_amd.models import simulate
wl = bench.WORKLOADS["occu_re"]
with contextlib.redirect_stdout(io.StringIO()):
    d, _ = simulate(**wl["cfg"])
ds = OccuDataset(d["site_covs"], d["obs_covs"], d["obs"], model=wl["model"], **wl["options"])
for k in [int(a) for a in sys.argv[1:]] or [32, 28, 24, 20, 16, 12]:
    os.environ["BIOLITH_HIP_RE_WGS"] = str(k)
    us = []
    for s in range(2):
        r = ds.nuts(num_warmup=300, num_samples=300, num_chains=4, seed=s)
        us.append(1e3 * r.kernel_ms / r.n_leapfrog.reshape(4, -1).sum(axis=1).max())
    print(f"k={k:3d} (ran {r.wgs_per_chain}): {us[1]:.3f} us/leapfrog of the slowest chain  lds {r.lds_bytes}  {r.kernel_name.strip()}  {r.env_overrides}", flush=True)
