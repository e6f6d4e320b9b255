import os, sys, hashlib
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np
from biolith_amd.engine import OccuDataset
from conftest import quiet_simulate
d = quiet_simulate(n_sites=1000, n_site_covs=3, n_obs_covs=3, deployment_days_per_site=35, session_duration=7)[0]
ds = OccuDataset(d["site_covs"], d["obs_covs"], d["obs"])
def h(r): return hashlib.sha256(r.draws.tobytes() + r.num_steps.tobytes()).hexdigest()[:16]
os.environ.pop("BIOLITH_HIP_GENERAL", None)
a = ds.nuts(num_warmup=150, num_samples=100, num_chains=2, seed=5)
os.environ["BIOLITH_HIP_GENERAL"] = "1"
b = ds.nuts(num_warmup=150, num_samples=100, num_chains=2, seed=5)
os.environ.pop("BIOLITH_HIP_GENERAL", None)
print(os.environ.get("BIOLITH_HIP_LIB"), a.kernel_name, h(a), "|", b.kernel_name, h(b), "same" if h(a) == h(b) else "DIFFERENT",
      "first differing transition", int(np.argmax((a.num_steps != b.num_steps).any(0))) if (a.num_steps != b.num_steps).any() else -1)
