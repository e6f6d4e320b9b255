#!/usr/bin/env python3
"""What each of occu_rn's 160 compute waves (32 workgroups x 5, BASELINE.json configs[3]) holds under the engine's contiguous deal of the
sites: sites, detections, items of 8 terms (the 20-nat rule at the simulating coefficients, as tools/rn_workload_stats.py), the largest
item count of a site.  With a stamps file from tools/stamps_rn_waves.py as argument: the correlation of the measured per-wave cycles
with those numbers (CPU only otherwise).   python tools/rn_wave_composition.py [gpurun_out/rn/stamps_waves.txt]"""
import contextlib, io, os, sys
import numpy as np
from scipy.special import gammaln
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from biolith_amd.models import simulate_rn  # noqa: E402

with contextlib.redirect_stdout(io.StringIO()):
    data, truth = simulate_rn(n_sites=5000, n_site_covs=3, n_obs_covs=3, deployment_days_per_site=70, session_duration=7, random_seed=0)
Y = data["obs"][0, :, 0, :]
d = np.nansum(Y, axis=1).astype(int)
beta, alpha = np.asarray(truth["beta"]).reshape(-1), np.asarray(truth["alpha"]).reshape(-1)
X, W = data["site_covs"], data["obs_covs"][:, 0]
n = np.arange(0, 101)
eta, nu = beta[0] + X @ beta[1:], alpha[0] + W @ alpha[1:]
lq = -np.logaddexp(0.0, nu)
a = eta + (lq * (Y == 0)).sum(1)
L = n[None, :] * a[:, None] - gammaln(n + 1)[None, :]
with np.errstate(divide="ignore"):
    L = L + np.where((Y == 1)[:, :, None], np.log1p(-np.exp(lq[:, :, None] * n[None, None, :])), 0).sum(1)
cut = np.array([np.max(np.nonzero(x)[0]) for x in (L >= L.max(1)[:, None] - 20.0)])
nch = np.ceil(cut / 8).clip(1).astype(int)
k, cw, N = 32, 5, 5000
nloc = -(-N // k)
rows = []
for b in range(k):
    s0, cnt = b * nloc, min(nloc, N - b * nloc)
    spw = -(-cnt // cw)
    for w in range(cw):
        lo, hi = s0 + w * spw, s0 + min(cnt, (w + 1) * spw)
        rows.append((b, w, hi - lo, int(nch[lo:hi].sum()), int(nch[lo:hi].max()), int((d[lo:hi] >= 9).sum()), int((nch[lo:hi] >= 3).sum())))
rows = np.array(rows)
print("per wave: sites", np.bincount(rows[:, 2])[28:].tolist(), "(from 28);  items min / mean / max", rows[:, 3].min(), rows[:, 3].mean().round(1), rows[:, 3].max(),
      "; waves by largest item count of a site (1..):", np.bincount(rows[:, 4])[1:].tolist(), "; waves with a site of >= 3 items:", int((rows[:, 6] > 0).sum()))
if len(sys.argv) > 1:
    cyc = []
    for ln in open(sys.argv[1]):
        p = ln.split()
        if len(p) == 1 + cw and p[0].isdigit():
            cyc += [float(x) for x in p[1:]]
    cyc = np.array(cyc)
    assert len(cyc) == len(rows)
    print(f"measured per-wave cycles: min {cyc.min():.0f} mean {cyc.mean():.0f} max {cyc.max():.0f} sd {cyc.std():.0f}")
    for name, col in (("sites", 2), ("items", 3), ("largest item count", 4), ("sites with >= 9 detections", 5), ("sites with >= 3 items", 6)):
        print(f"  correlation with {name}: {np.corrcoef(cyc, rows[:, col])[0, 1]:+.2f}")
    for m in sorted(set(rows[:, 4])):
        s = rows[:, 4] == m
        print(f"  waves whose largest site has {m} items: {int(s.sum()):3d}, mean cycles {cyc[s].mean():.0f}")
    full = rows[:, 2] >= 32
    print(f"  waves of 32 sites: {int(full.sum())}, mean {cyc[full].mean():.0f}; of fewer: {int((~full).sum())}, mean {cyc[~full].mean():.0f}")
