#!/usr/bin/env python3
"""Per-compute-wave evaluation cycles and the tick's parts for a row of the reference's benchmark grid (benchmarks/occu_spoccupancy.py:16-70;
one chain, 100 + 500; 3 + 3 covariates instead of the grid's 2 + 1: the stamps library is built for that capacity) on the stamps library:   python tools/stamps_waves_grid.py 6 7"""
import contextlib, io, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("BIOLITH_HIP_LIB", os.path.join(ROOT, "biolith_amd", "lib", "libbiolith_hip_stamps.so"))
from biolith_amd.engine import OccuDataset
from biolith_amd.models import simulate
for i in [int(a) for a in sys.argv[1:]] or [6, 7]:
    n_sites, visits = int(100 * 2 ** i), int(8 * 2 ** (i / 2))
    with contextlib.redirect_stdout(io.StringIO()):
        data, _ = simulate(n_site_covs=3, n_obs_covs=3, n_sites=n_sites, deployment_days_per_site=visits * 7, session_duration=7, simulate_missing=False, random_seed=42 + i)
    ds = OccuDataset(data["site_covs"], data["obs_covs"], data["obs"])
    r = ds.nuts(num_warmup=100, num_samples=500, num_chains=1, seed=0)
    c = ds.debug_counters(544)
    passes = max(int(c[20]), 1); k = r.wgs_per_chain; cw = r.threads_per_wg // 64 - 1
    w = c[32:32 + 8 * min(k, 32)].reshape(min(k, 32), 8)[:, :cw].astype(np.float64) / passes
    ticks = int(c[8]); tot = c[:7].sum()
    print(f"row {i}: {n_sites} x {visits}: {r.kernel_name.strip()} k={k} lanes/pair {r.lane_group} compute waves {cw}; {1e3 * r.kernel_ms / (r.n_leapfrog.sum() + 1):.2f} us/leapfrog (stamps build)")
    print(f"   per-wave evaluation cycles (first {min(k, 32)} workgroups): min {w.min():.0f} mean {w.mean():.0f} max {w.max():.0f}; by wave index {[round(x) for x in w.mean(0)]}")
    print(f"   cycles per tick {tot / max(ticks, 1):.0f}: " + ", ".join(f"{n} {v / max(ticks, 1):.0f}" for n, v in zip(["decide", "wait compute", "publish", "poll", "spec", "barrier2"], c[:6])))
    ds.close()
