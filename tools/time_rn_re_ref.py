import sys, time, contextlib, io
sys.path.insert(0,'.'); sys.path.insert(0,'tests')
import numpy as np
from biolith_amd.models import occu_rn, simulate_rn
from biolith_amd.utils import fit
with contextlib.redirect_stdout(io.StringIO()):
    data, truth = simulate_rn(simulate_missing=True)
for kw in (dict(obs_random_effects=True), dict(site_random_effects=True)):
    t=time.time(); res = fit(occu_rn, **data, **kw, num_chains=1, num_warmup=300, num_samples=200, timeout=600); w=time.time()-t
    r=res.mcmc.result
    print(kw, f"wall {w:.1f}s kernel {r.kernel_ms:.0f} ms steps/transition {r.num_steps.mean():.1f} leapfrogs {int(r.n_leapfrog.sum())} us/leapfrog {1e3*r.kernel_ms/r.n_leapfrog.sum():.1f} k={r.wgs_per_chain} {r.kernel_name.strip()} lds {r.lds_bytes} staged {r.lds_staged} env '{r.env_overrides}' div {int(r.diverging.sum())}")
