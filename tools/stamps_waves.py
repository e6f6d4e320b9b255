#!/usr/bin/env python3
"""Chain 0's site-evaluation cycles PER COMPUTE WAVE on the stamps library, for a bench workload:
    python tools/stamps_waves.py headline|stacked|dyn|rn          (tools/stamps_rn_waves.py is the occu_rn form with its table)
The tick waits for the slowest wave of the chain; a wave index that is slow in every workgroup is a SIMD shared with another wave."""
import contextlib, io, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("BIOLITH_HIP_LIB", os.path.join(ROOT, "biolith_amd", "lib", "libbiolith_hip_stamps.so"))
import bench
from biolith_amd.engine import OccuDataset
from biolith_amd.models import simulate, simulate_dyn, simulate_rn
which = sys.argv[1] if len(sys.argv) > 1 else "headline"
wl = bench.WORKLOADS[{"headline": "occu", "stacked": "occu_stacked", "dyn": "occu_dyn", "rn": "occu_rn"}[which]]
with contextlib.redirect_stdout(io.StringIO()):
    d, _ = {"occu_rn": simulate_rn, "occu_dyn": simulate_dyn}.get(wl["model"], simulate)(**wl["cfg"])
ds = OccuDataset(d["site_covs"], d["obs_covs"], d["obs"], model=wl["model"])
r = ds.nuts(num_warmup=300, num_samples=300, num_chains=4, seed=0)
c = ds.debug_counters(544)
passes = max(int(c[20]), 1)
k, cw = r.wgs_per_chain, r.threads_per_wg // 64 - 1
w = c[32:32 + 8 * min(k, 64)].reshape(min(k, 64), 8)[:, :cw].astype(np.float64) / passes
print(f"{which}: {r.kernel_name.strip()} k={k} compute waves {cw} passes {passes}; per-wave site-evaluation cycles: min {w.min():.0f} mean {w.mean():.0f} max {w.max():.0f}; "
      f"by wave index (mean over workgroups): {[round(x) for x in w.mean(0)]}; per-workgroup max: mean {w.max(1).mean():.0f} max {w.max(1).max():.0f}")
ticks = int(c[8]); tot = c[:7].sum()
print(f"   cycles per tick {tot / max(ticks, 1):.0f}: " + ", ".join(f"{n} {v / max(ticks, 1):.0f}" for n, v in zip(["decide", "wait compute", "publish", "poll", "spec", "barrier2"], c[:6])))
