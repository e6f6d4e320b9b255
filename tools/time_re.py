#!/usr/bin/env python3
"""Per-leapfrog time of the random-effects occupancy model (re_kernel.hpp) on the reference's test shapes and a larger one."""
import contextlib, io, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from biolith_amd.engine import OccuDataset
from biolith_amd.models import simulate

def run(name, data, chains=4, **kw):
    ds = OccuDataset(data["site_covs"], data["obs_covs"], data["obs"], model="occu_re", **kw)
    r = ds.nuts(num_warmup=300, num_samples=300, num_chains=chains, seed=0)
    r = ds.nuts(num_warmup=300, num_samples=300, num_chains=chains, seed=1)
    # the launch ends with its slowest chain: over the MEAN leapfrogs per chain (bench.py's figure) the time also moves with how evenly
    # the chains adapted; over the slowest chain's leapfrogs it is the kernel's own
    per_chain = r.n_leapfrog.reshape(chains, -1).sum(axis=1)
    print(f"{name:52s} D={ds.D:6d} chains={chains} kernel {r.kernel_ms:9.2f} ms  {1e3 * r.kernel_ms / per_chain.mean():8.2f} us/leapfrog/chain  "
          f"({1e3 * r.kernel_ms / per_chain.max():6.2f} over the slowest chain's)  steps/transition {r.num_steps.mean():6.1f} div {r.diverging.mean():.3f}")

with contextlib.redirect_stdout(io.StringIO()):
    d_site, _ = simulate(site_random_effects=True, deployment_days_per_site=7000)
    d_obs, _ = simulate(simulate_missing=True)
    d_big, _ = simulate(n_sites=2000, n_site_covs=3, n_obs_covs=3, deployment_days_per_site=70, session_duration=7, site_random_effects=True)
run("site effects, 100 sites x 1000 visits (occu.py:770)", d_site, site_random_effects=True)
run("obs effects, 100 sites x 52 visits (occu.py:823)", d_obs, obs_random_effects=True)
run("both, 100 sites x 52 visits (occu.py:843)", d_obs, site_random_effects=True, obs_random_effects=True)
run("site effects, 2000 sites x 10 visits", d_big, site_random_effects=True)
run("both, 2000 sites x 10 visits", d_big, site_random_effects=True, obs_random_effects=True)
