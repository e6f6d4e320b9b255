#!/usr/bin/env python3
"""What BASELINE.json configs[3] (occu_rn, 5 000 sites x 10 visits) asks of the Royle-Nichols evaluator, counted in NumPy at the simulating
coefficients and at six points around them (CPU only):  detections per site, the n-range a site needs (terms within 20 nats of its
largest, the kernel's rule), items of 8 terms per site, how stable a site's range is across posterior-like points, and the chunk
counts by detection count.   python tools/rn_workload_stats.py > profiles/r05/k_rn_workload_stats.txt"""
import contextlib
import io
import os
import sys

import numpy as np
from scipy.special import gammaln

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from biolith_amd.models import simulate_rn  # noqa: E402

with contextlib.redirect_stdout(io.StringIO()):
    data, truth = simulate_rn(n_sites=5000, n_site_covs=3, n_obs_covs=3, deployment_days_per_site=70, session_duration=7, random_seed=0)
Y = data["obs"][0, :, 0, :]
d = np.nansum(Y, axis=1).astype(int)
beta0, alpha0 = np.asarray(truth["beta"]).reshape(-1), np.asarray(truth["alpha"]).reshape(-1)
X, W = data["site_covs"], data["obs_covs"][:, 0]
n = np.arange(0, 101)


def cutoff(beta, alpha, nats=20.0):
    eta, nu = beta[0] + X @ beta[1:], alpha[0] + W @ alpha[1:]
    q = 1.0 - 1.0 / (1.0 + np.exp(-nu))
    a = eta + (np.log(q) * (Y == 0)).sum(1)
    L = n[None, :] * a[:, None] - gammaln(n + 1)[None, :]
    with np.errstate(divide="ignore"):
        L = L + np.where((Y == 1)[:, :, None], np.log1p(-(q[:, :, None] ** n[None, None, :])), 0).sum(1)
    return np.array([np.max(np.nonzero(x)[0]) for x in (L >= L.max(1)[:, None] - nats)])


print("detections per site (share of the 5 000 sites), 0 .. 10:", np.round(np.bincount(d, minlength=11) / 5000, 4).tolist(), "mean", d.mean())
rng = np.random.default_rng(0)
cuts = [cutoff(beta0, alpha0)] + [cutoff(beta0 + rng.normal(size=4) * 0.03, alpha0 + rng.normal(size=4) * 0.03) for _ in range(6)]
c0 = cuts[0]
ch = np.ceil(c0 / 8).clip(1)
print(f"at the simulating coefficients: mean largest n needed {c0.mean():.2f}; items of 8 terms per site {ch.mean():.3f} (sites with detections: {ch[d > 0].mean():.3f});"
      f" items in all {int(ch.sum())}, of sites with detections {int(ch[d > 0].sum())}; item x visit-pairs if only detections' pairs ran {int((ch * np.ceil(d / 2)).sum())} against {int(ch.sum() * 5)} now")
print("items per site, share of sites (1, 2, 3, 4, 5 items):", np.round(np.bincount(ch.astype(int))[1:] / 5000, 4).tolist())
cmax, cmin = np.max(cuts, axis=0), np.min(cuts, axis=0)
print("largest n needed, max - min over the 7 points, share of sites (0, 1, 2, ...):", np.round(np.bincount((cmax - cmin).astype(int)) / 5000, 4).tolist())
print("by detections: sites, mean largest n (max over the 7 points), share needing n > 16, > 24, > 32")
for k in range(11):
    s = d == k
    print(f"  {k:2d} {int(s.sum()):5d} {cmax[s].mean():6.1f} {np.mean(cmax[s] > 16):7.3f} {np.mean(cmax[s] > 24):7.4f} {np.mean(cmax[s] > 32):7.4f}")
