#!/usr/bin/env python3
"""Soak run of the exchange paths (not a test: a few minutes of launches over shapes, chain counts and geometries; any timeout of an
exchange surfaces as BL_ERR_TIMEOUT)."""
import contextlib, io, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from biolith_amd.engine import OccuDataset
from biolith_amd.models import simulate
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 240.0
t0, n, leap = time.time(), 0, 0
rng = np.random.default_rng(0)
sets = []
with contextlib.redirect_stdout(io.StringIO()):
    for (ns, days) in ((10000, 35), (3000, 70), (700, 35), (12800, 630)):
        d, _ = simulate(n_sites=ns, n_site_covs=3, n_obs_covs=3, deployment_days_per_site=days, session_duration=7)
        sets.append(("occu", OccuDataset(d["site_covs"], d["obs_covs"], d["obs"])))
    d, _ = simulate(n_sites=2000, n_site_covs=3, n_obs_covs=3, deployment_days_per_site=70, session_duration=7, site_random_effects=True)
    sets.append(("occu_re", OccuDataset(d["site_covs"], d["obs_covs"], d["obs"], model="occu_re", site_random_effects=True)))
    d, _ = simulate(n_species=2, n_sites=1000, n_site_covs=3, n_obs_covs=3, deployment_days_per_site=70, session_duration=7, site_random_effects=True)
    sets.append(("occu_re 2 species", OccuDataset(d["site_covs"], d["obs_covs"], d["obs"], model="occu_re", site_random_effects=True)))
    from biolith_amd.models import simulate_rn
    d, _ = simulate_rn(n_sites=5000, n_site_covs=3, n_obs_covs=3, deployment_days_per_site=70, session_duration=7)
    sets.append(("occu_rn", OccuDataset(d["site_covs"], d["obs_covs"], d["obs"], model="occu_rn")))      # config 4: speculation behind even leaves
    d, _ = simulate(n_sites=2000, n_periods=8, n_site_covs=3, n_obs_covs=3, deployment_days_per_site=28, session_duration=7)
    sets.append(("occu", OccuDataset(d["site_covs"], d["obs_covs"], d["obs"])))                           # 2 000 x 8 x 4: one period per lane
    sets.append(("occu_dyn", OccuDataset(d["site_covs"], d["obs_covs"], d["obs"], model="occu_dyn")))    # the two-scans form
while time.time() - t0 < budget:
    name, ds = sets[n % len(sets)]
    C = int(rng.choice([1, 2, 4, 8, 16] if name == "occu" else [1, 2, 4]))
    r = ds.nuts(num_warmup=int(rng.choice([50, 200, 600])), num_samples=int(rng.choice([50, 300])), num_chains=C, seed=int(rng.integers(1 << 30)))
    assert np.all(np.isfinite(r.draws)), (name, C)
    n += 1
    leap += int(r.n_leapfrog.sum())
print(f"soak: {len(sets)} datasets, {n} launches, {leap} leapfrogs, {time.time() - t0:.0f} s, no exchange timed out")
