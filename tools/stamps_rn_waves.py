#!/usr/bin/env python3
"""occu_rn at BASELINE.json's config 4 on the stamps library: chain 0's site-evaluation cycles PER COMPUTE WAVE (32 workgroups x 5 waves),
summed over the run -- how uneven the waves' streams are (the tick waits for the slowest of 160), and what the sites of a wave have to do
with it (detections, items at the posterior mean).   python tools/stamps_rn_waves.py > gpurun_out/rn/stamps_waves.txt"""
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
c = ds.debug_counters(544)
passes = max(int(c[20]), 1)
k, cw = r.wgs_per_chain, 5
w = c[32:32 + 8 * k].reshape(k, 8)[:, :cw].astype(np.float64) / passes
print(f"k={k} passes {passes}; per-wave site-evaluation cycles (mean over passes): min {w.min():.0f} mean {w.mean():.0f} max {w.max():.0f}; "
      f"per workgroup max: min {w.max(1).min():.0f} mean {w.max(1).mean():.0f} max {w.max(1).max():.0f}")
print("per workgroup (rows) x wave:")
for b in range(k):
    print("  ", b, " ".join(f"{x:6.0f}" for x in w[b]))
hw = c[32 + 256:32 + 256 + 8 * k].reshape(k, 8)
print("SIMD of [compute waves 0..4 | control wave] per workgroup (HW_ID bits 5:4), and which compute waves share one:")
for b in range(k):
    simd = [(int(x) >> 4) & 3 for x in list(hw[b, :cw]) + [hw[b, 7]]]
    pairs = [(i, j) for i in range(cw + 1) for j in range(i + 1, cw + 1) if simd[i] == simd[j]]
    print("  ", b, simd, " shared:", [("c%d" % i if i < cw else "ctl", "c%d" % j if j < cw else "ctl") for i, j in pairs], " cycles", " ".join(f"{x:.0f}" for x in w[b]))
