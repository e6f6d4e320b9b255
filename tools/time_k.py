#!/usr/bin/env python3
"""Headline workload at several workgroups-per-chain settings:  python tools/time_k.py [k ...]   (BIOLITH_HIP_LIB / BIOLITH_HIP_CWAVES select a variant build and its compute-wave count)."""
import contextlib, io, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from biolith_amd.engine import OccuDataset
from biolith_amd.models import simulate
with contextlib.redirect_stdout(io.StringIO()):
    d, _ = simulate(n_sites=10000, n_site_covs=3, n_obs_covs=3, deployment_days_per_site=35, session_duration=7)
ds = OccuDataset(d["site_covs"], d["obs_covs"], d["obs"])
ks = [int(x) for x in sys.argv[1:]] or [0, 20, 24, 27, 29, 32]
for k in ks:
    ms = []
    for s in range(5):
        r = ds.nuts(num_warmup=1000, num_samples=1000, num_chains=4, seed=s, wgs_per_chain=k)
        ms.append(r.kernel_ms)
    print(f"wgs_per_chain={k:2d} -> k={r.wgs_per_chain} threads {r.threads_per_wg}: kernel ms {np.round(ms[1:], 2).tolist()} mean {np.mean(ms[1:]):.2f}")
