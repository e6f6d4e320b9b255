#!/usr/bin/env python3
"""Section cycles of one leapfrog of the random-effects kernel (BL_STAMPS build: make -C biolith_amd/csrc stamps)."""
import contextlib, io, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("BIOLITH_HIP_LIB", os.path.join(ROOT, "biolith_amd", "lib", "libbiolith_hip_stamps.so"))
from biolith_amd.engine import OccuDataset
from biolith_amd.models import simulate
with contextlib.redirect_stdout(io.StringIO()):
    d_obs, _ = simulate(simulate_missing=True, n_site_covs=3, n_obs_covs=3)   # (the stamps build holds the (3,3) kernels only)
    d_big, _ = simulate(n_sites=2000, n_site_covs=3, n_obs_covs=3, deployment_days_per_site=70, session_duration=7, site_random_effects=True)
    d_bench, _ = simulate(n_sites=10000, n_site_covs=3, n_obs_covs=3, deployment_days_per_site=70, session_duration=7, site_random_effects=True)
NAMES = ["effect squares, prior", "exchange 1 tail: fixed-effect gradients, potential", "second half step + dot products", "exchange 2",
         "decisions", "vector decisions + next half step", "new transition", "barrier at leaf start", "site pass", "block sum 1", "exchange 1",
         "block sum 2"]
for name, data, kw in (("both, 100 x 52", d_obs, dict(site_random_effects=True, obs_random_effects=True)),
                       ("site effects, 2000 x 10", d_big, dict(site_random_effects=True)),
                       ("both, 2000 x 10", d_big, dict(site_random_effects=True, obs_random_effects=True)),
                       ("site effects, 10000 x 10 (bench)", d_bench, dict(site_random_effects=True))):
    ds = OccuDataset(data["site_covs"], data["obs_covs"], data["obs"], model="occu_re", **kw)
    r = ds.nuts(num_warmup=200, num_samples=200, num_chains=4, seed=0)
    c = ds.debug_counters()
    n = max(int(c[16]), 1)
    print(f"{name}: D={ds.D} {1e3 * r.kernel_ms / (r.n_leapfrog.sum() / 4):.2f} us/leapfrog, {c[:12].sum() / n:.0f} cycles")
    for nm, v in zip(NAMES, c[:12]):
        print(f"    {nm:40s} {v / n:8.0f} cyc")
