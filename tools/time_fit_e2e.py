#!/usr/bin/env python3
"""Wall time of the whole fit() call at the headline size (what biolith's own benchmark times), split into its parts."""
import contextlib, io, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from biolith_amd.engine import OccuDataset
from biolith_amd.models import occu, simulate
from biolith_amd.utils import fit
with contextlib.redirect_stdout(io.StringIO()):
    d, _ = simulate(n_sites=10000, n_site_covs=3, n_obs_covs=3, deployment_days_per_site=35, session_duration=7)
fit(occu, **d, num_chains=4, num_samples=10, num_warmup=10)  # library load, first-touch
for rep in range(3):
    t0 = time.perf_counter()
    r = fit(occu, **d, num_chains=4)
    t1 = time.perf_counter()
    m = float(r.samples["psi"].mean())
    t2 = time.perf_counter()
    print(f"fit() {1e3 * (t1 - t0):.1f} ms (kernel {r.mcmc.result.kernel_ms:.1f} ms), psi.mean() {1e3 * (t2 - t1):.1f} ms  -> {m:.4f}")
t0 = time.perf_counter(); ds = OccuDataset(d["site_covs"], d["obs_covs"], d["obs"]); t1 = time.perf_counter()
res = ds.nuts(num_warmup=1000, num_samples=1000, num_chains=4); t2 = time.perf_counter()
psi = ds.deterministic(res.draws.reshape(-1, ds.D), psi=True)[0]; t3 = time.perf_counter()
print(f"dataset {1e3 * (t1 - t0):.1f} ms, nuts {1e3 * (t2 - t1):.1f} ms, psi (4000 x 10000) {1e3 * (t3 - t2):.1f} ms")
