#!/usr/bin/env python3
"""Kernel time of the headline workload with the production library (median of several launches)."""
import contextlib, io, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from biolith_amd.engine import OccuDataset
from biolith_amd.models import simulate
with contextlib.redirect_stdout(io.StringIO()):
    d, _ = simulate(n_sites=10000, n_site_covs=3, n_obs_covs=3, deployment_days_per_site=35, session_duration=7)
ds = OccuDataset(d["site_covs"], d["obs_covs"], d["obs"])
n = int(sys.argv[1]) if len(sys.argv) > 1 else 7
k = int(sys.argv[2]) if len(sys.argv) > 2 else 0  # workgroups per chain (0 = the engine's choice)
ms, us = [], []
for t in range(n):
    r = ds.nuts(num_warmup=1000, num_samples=1000, num_chains=4, seed=t, wgs_per_chain=k)
    ms.append(r.kernel_ms); us.append(r.kernel_ms * 1e3 / (r.n_leapfrog.sum() / 4))
print(f"kernel ms median {np.median(ms):.2f} min {np.min(ms):.2f}  us/leapfrog/chain median {np.median(us):.3f}  k={r.wgs_per_chain} l2local={r.chains_l2_local}")
