#!/usr/bin/env python3
"""Per-leapfrog time of random effects with several species under one chain (shared sds)."""
import contextlib, io, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from biolith_amd.engine import OccuDataset
from biolith_amd.models import simulate
for S, n in ((2, 2000), (3, 2000), (4, 1000)):
    with contextlib.redirect_stdout(io.StringIO()):
        d, _ = simulate(n_species=S, n_sites=n, n_site_covs=3, n_obs_covs=3, deployment_days_per_site=70, session_duration=7, site_random_effects=True)
    ds = OccuDataset(d["site_covs"], d["obs_covs"], d["obs"], model="occu_re", site_random_effects=True)
    r = ds.nuts(num_warmup=300, num_samples=300, num_chains=4, seed=0)
    r = ds.nuts(num_warmup=300, num_samples=300, num_chains=4, seed=1)
    print(f"{S} species x {n} sites x 10 visits, site effects: D={ds.D} k={r.wgs_per_chain} kernel {r.kernel_ms:8.2f} ms "
          f"{1e3 * r.kernel_ms / (r.n_leapfrog.sum() / 4):7.2f} us/leapfrog/chain  div {r.diverging.mean():.3f}")
