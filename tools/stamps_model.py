#!/usr/bin/env python3
"""tools/stamps.py for one of the secondary models of tools/time_models.py:  python tools/stamps_model.py occu_cop|occu_fp|nmixture|occu|occu_dyn|occu_stacked"""
import contextlib, io, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("BIOLITH_HIP_LIB", os.path.join(ROOT, "biolith_amd", "lib", "libbiolith_hip_stamps.so"))
from biolith_amd.engine import OccuDataset
from biolith_amd.models import simulate, simulate_cop, simulate_dyn, simulate_nmixture
which = sys.argv[1] if len(sys.argv) > 1 else "occu_cop"
kw = dict(n_sites=5000, n_site_covs=3, n_obs_covs=3, deployment_days_per_site=70, session_duration=7)
with contextlib.redirect_stdout(io.StringIO()):
    if which == "occu_cop":
        d, _ = simulate_cop(**kw)
        ds = OccuDataset(d["site_covs"], d["obs_covs"], d["obs"], model="occu_cop", fp_mode="constant", session_duration=d["session_duration"])
    elif which == "occu_fp":
        d, _ = simulate(**kw, prob_fp_constant=0.1)
        ds = OccuDataset(d["site_covs"], d["obs_covs"], d["obs"], model="occu_fp", fp_mode="constant")
    elif which == "nmixture":
        d, _ = simulate_nmixture(**kw, min_abundance=0.5, max_abundance=8.0, max_observation_rate=6.0)
        ds = OccuDataset(d["site_covs"], d["obs_covs"], d["obs"], model="nmixture", max_abundance=100)
    elif which == "occu_dyn":
        d, _ = simulate_dyn(n_sites=2000, n_periods=8, n_site_covs=3, n_obs_covs=3, deployment_days_per_site=28, session_duration=7)
        ds = OccuDataset(d["site_covs"], d["obs_covs"], d["obs"], model="occu_dyn")
    elif which == "occu_small":   # one workgroup per chain (k = 1: no exchange), 7 compute waves
        d, _ = simulate(n_sites=200, n_site_covs=3, n_obs_covs=3, deployment_days_per_site=70, session_duration=7)
        ds = OccuDataset(d["site_covs"], d["obs_covs"], d["obs"])
    elif which == "occu_cfg1like":   # simulate()'s defaults' shape (100 x 52) at the stamps build's 3 + 3 covariates
        d, _ = simulate(n_sites=100, n_site_covs=3, n_obs_covs=3)
        ds = OccuDataset(d["site_covs"], d["obs_covs"], d["obs"])
    elif which == "occu_stacked":
        d, _ = simulate(n_sites=2000, n_periods=8, n_site_covs=3, n_obs_covs=3, deployment_days_per_site=28, session_duration=7)
        ds = OccuDataset(d["site_covs"], d["obs_covs"], d["obs"])
    else:
        d, _ = simulate(**kw)
        ds = OccuDataset(d["site_covs"], d["obs_covs"], d["obs"])
r = ds.nuts(num_warmup=300, num_samples=300, num_chains=4, seed=1)
c = ds.debug_counters()
ticks = int(c[8]); tot = c[:7].sum()
print(f"{which}: k={r.wgs_per_chain} threads {r.threads_per_wg} lanes/pair {r.lane_group} kernel {r.kernel_ms:.1f} ms, exchanges {ticks}, leapfrogs/chain {r.n_leapfrog.sum() / 4:.0f}, cycles/exchange {tot / max(ticks, 1):.0f}, "
      f"mean steps/transition {r.num_steps.mean():.1f}, evaluations {int(c[20])} ({c[21] / max(float(c[20]), 1.0):.0f} cyc each)")
names = ["decisions+bookkeeping", "wait for compute", "publish", "sweep", "spec position", "barrier2"]
print("   " + " | ".join(f"{n} {v / max(ticks, 1):.0f}" for n, v in zip(names, c[:6])))
kn = c[11:14].astype(float); kc = np.array([c[14], c[15], c[7]], dtype=float)
print("   " + " | ".join(f"{n}: {int(a)} x {b / max(a, 1):.0f}" for n, a, b in zip(("leaf", "doubling", "end"), kn, kc)))
