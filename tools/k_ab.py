"""Headline workload: us per leapfrog of the slowest chain at several workgroups-per-chain settings, 12 seeds each, interleaved (kernel ms alone
misleads: another k is another summation order, other trees, another leapfrog count).   python tools/k_ab.py"""
import contextlib, io, os, sys
import numpy as np
sys.path.insert(0, "/root/repo")
from biolith_amd.engine import OccuDataset
from biolith_amd.models import simulate
with contextlib.redirect_stdout(io.StringIO()):
    d, _ = simulate(n_sites=10000, n_site_covs=3, n_obs_covs=3, deployment_days_per_site=35, session_duration=7)
ds = OccuDataset(d["site_covs"], d["obs_covs"], d["obs"])
res = {27: [], 29: [], 32: []}
for s in range(14):
    for k in res:
        r = ds.nuts(num_warmup=1000, num_samples=1000, num_chains=4, seed=s, wgs_per_chain=k)
        if s >= 2: res[k].append(1e3 * r.kernel_ms / r.n_leapfrog.sum(axis=1).max())
for k, v in res.items():
    print(f"k={k}: us per leapfrog of the slowest chain, 12 seeds: mean {np.mean(v):.4f} sd {np.std(v):.4f} min {np.min(v):.4f} max {np.max(v):.4f}")
