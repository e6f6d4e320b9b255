import contextlib, io, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("BIOLITH_HIP_LIB", os.path.join(ROOT, "biolith_amd", "lib", "libbiolith_hip_stamps.so"))
from biolith_amd.engine import OccuDataset
from biolith_amd.models import simulate_rn
with contextlib.redirect_stdout(io.StringIO()):
    drn, _ = simulate_rn(n_sites=5000, n_site_covs=3, n_obs_covs=3, deployment_days_per_site=70, session_duration=7)
ds = OccuDataset(drn["site_covs"], drn["obs_covs"], drn["obs"], model="occu_rn")
r = ds.nuts(num_warmup=300, num_samples=300, num_chains=4, seed=0)
c = ds.debug_counters()
ticks, rt = int(c[8]), int(c[9])
tot = c[:7].sum()
print(f"k={r.wgs_per_chain} kernel {r.kernel_ms:.1f} ms ticks {ticks} cycles/tick {tot / max(ticks, 1):.0f} us/tick {r.kernel_ms * 1e3 / max(ticks, 1):.2f}")
for n, v in zip(["decide", "wait compute", "wg partial+publish", "sweep(poll)", "spec", "barrier2"], c[:6]):
    print(f"    {n:30s} {v / max(ticks, 1):8.0f} cyc  {100.0 * v / tot:5.1f} %")
print(f"    site evaluation (compute wave 1): {int(c[20])} passes, {c[21] / max(float(c[20]), 1.0):.0f} cyc each")
names=["x, eta, prior","A0 visits (lane = site)","bounds, cutoff bisection, records, scan","item map + record reads","LP init + A1","floors + B + combine","floor grads + C","site results"]
n=max(int(c[20]),1)
for i,nm in enumerate(names): print(f"      {nm:18s} {c[22+i]/n:8.0f} cyc")
print("      sum", c[22:30].sum()/n)
