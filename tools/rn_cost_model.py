#!/usr/bin/env python3
"""Cost model of occu_rn's site evaluation at config 4 (CPU only): the kernel's per-site n-cutoffs (rn_device.hpp: last n whose
upper bound g(n) = max(n a - lgamma(n+1), p_n + n* cnon) is within 20 nats of the lower bound of the best term) at the true
parameters, the items (site, 8-term chunk) they make, and how those fall onto a chain's waves (k workgroups x 7 waves,
contiguous sites, whole sites packed into rounds of <= 64 lanes)."""
import contextlib, io, os, sys
import numpy as np
from scipy.special import gammaln
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from biolith_amd.models import simulate_rn

with contextlib.redirect_stdout(io.StringIO()):
    data, truth = simulate_rn(**bench.CFG4)
X, W, Y = data["site_covs"], data["obs_covs"][:, 0], data["obs"][0, :, 0]          # (N,3) (N,J,3) (N,J)
beta, alpha = truth["beta"][0], truth["alpha"][0]
scale = float(sys.argv[1]) if len(sys.argv) > 1 else 1.0   # scale != 1: away from the truth (warmup-like)
beta, alpha = beta * scale, alpha * scale
N, J = Y.shape
K, FL = 100, -15.942385
eta = beta[0] + X @ beta[1:]
nu = alpha[0] + W @ alpha[1:]
r = 1 / (1 + np.exp(-nu)); q = 1 - r
n = np.arange(K + 1)[None, :]
logq, logr = np.log(q), np.log(r)
det = (Y == 1); non = (Y == 0)
cnon = (logq * non).sum(1); clr = (logr * det).sum(1); ndet = det.sum(1)
lqmin = np.where(non, logq, 0.0).min(1)
nstar = np.where(lqmin < 0, FL / np.minimum(lqmin, -1e-30), 0.0)
a = eta + cnon
pn = n * eta[:, None] - gammaln(n + 1)
A = n * a[:, None] - gammaln(n + 1)
B = pn + (cnon * nstar)[:, None]
g = np.maximum(A, B)
def mode_lb(a):
    n1 = np.clip(np.floor(np.exp(a)), 1, K)
    return n1 * a - ((n1 + 0.5) * np.log(n1) + 1 - n1)
m_lb = np.maximum(np.maximum(ndet * -87.33654475, a + clr), mode_lb(a) + clr)
ok = g[:, 1:] >= (m_lb - 20)[:, None]
Kw = np.maximum(1, np.where(ok.any(1), K - np.argmax(ok[:, ::-1], axis=1), 1))
for CH in (4, 8, 16):
    nch = (Kw + CH - 1) // CH
    print(f"chunk {CH:2d}: cutoff mean {Kw.mean():.1f} median {np.median(Kw):.0f} p90 {np.percentile(Kw, 90):.0f} max {Kw.max()};  items per site mean {nch.mean():.2f}, "
          f"sites with 1 / 2 / 3+ items: {np.mean(nch == 1):.2f} / {np.mean(nch == 2):.2f} / {np.mean(nch >= 3):.2f}; terms evaluated per site {CH * nch.mean():.1f}")
    for k, CW in ((32, 7), (32, 3), (27, 3), (64, 3), (64, 7)):
        nloc = (N + k - 1) // k
        rounds, items = [], []
        for m in range(k):
            cnt = max(0, min(nloc, N - m * nloc))
            spw = (cnt + CW - 1) // CW
            for w in range(CW):
                s = nch[m * nloc + w * spw: m * nloc + min(cnt, (w + 1) * spw)]
                nr, cur = 0, 0
                for i0 in range(0, len(s), 64):
                    cur = 0; nr += 1 if len(s[i0:i0 + 64]) else 0
                    for x in s[i0:i0 + 64]:
                        if cur + x > 64: nr += 1; cur = 0
                        cur += x
                rounds.append(nr); items.append(int(s.sum()))
        rounds, items = np.array(rounds), np.array(items)
        print(f"   k={k:2d} x {CW} waves: sites per wave {nloc / CW:5.1f}; items per wave mean {items.mean():5.1f} max {items.max():3d}; item rounds per wave mean {rounds.mean():.2f} max {rounds.max()}"
              f"  ({np.mean(rounds > 1) * 100:.0f} % of the waves need more than one)")
