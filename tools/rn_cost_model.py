#!/usr/bin/env python3
"""Cost model of occu_rn's site evaluation at config 4 (CPU only): per-site n-cutoffs at the true parameters, and what the slowest
wave costs under the kernel's site -> wave assignment and under alternatives (sites sorted by a cutoff proxy, waves given rounds of
64 sorted sites so that their summed round-maxima balance)."""
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
N, J = Y.shape
K = 100
eta = beta[0] + X @ beta[1:]
nu = alpha[0] + W @ alpha[1:]
r = 1 / (1 + np.exp(-nu)); q = 1 - r
n = np.arange(K + 1)[None, :]
logq = np.log(q)
det = (Y == 1); non = (Y == 0)
cnon = (logq * non).sum(1)
pn = n * eta[:, None] - gammaln(n + 1)
with np.errstate(divide="ignore"):
    lb = np.where(det[:, :, None], np.log1p(-np.exp(n[None] * logq[:, :, None])), 0.0).sum(1)   # (N, K+1)
term = pn + n * cnon[:, None] + lb
term[:, 0] = np.where(det.any(1), -87.3 * det.sum(1), term[:, 0])
mx = term.max(1, keepdims=True)
cut_post = (term >= mx - 20).cumsum(1).argmax(1)          # last n within e^-20 of the best term (posterior sum)
prior_m = pn.max(1, keepdims=True)
cut_prior = (pn >= prior_m - 20).cumsum(1).argmax(1)
cut = np.maximum(cut_post, 1)
print(f"sites {N}: posterior cutoff mean {cut.mean():.1f} median {np.median(cut):.0f} p90 {np.percentile(cut, 90):.0f} max {cut.max()};"
      f" prior-sum cutoff mean {cut_prior.mean():.1f} max {cut_prior.max()}")

def slowest(order, waves, fixed=8.0):
    """rounds of 64 sites in `order`, dealt to `waves` waves in contiguous blocks balancing summed (fixed + round max)."""
    c = cut[order]
    rounds = [c[i:i + 64].max() + fixed for i in range(0, N, 64)]
    total = sum(rounds)
    # greedy contiguous partition
    target, loads, cur = total / waves, [], 0.0
    for x in rounds:
        if cur + x > target * 1.0 and cur > 0 and len(loads) < waves - 1:
            loads.append(cur); cur = 0.0
        cur += x
    loads.append(cur)
    return max(loads), total / waves, len(rounds)

waves = 81
ident = np.arange(N)
# the kernel today: 27 workgroups x 3 waves, one round each, contiguous sites
per_wave = [cut[i:i + 62].max() + 8.0 for i in range(0, N, 62)]
print(f"today (contiguous, one round of ~62 sites per wave, {len(per_wave)} waves): slowest {max(per_wave):.0f}, mean {np.mean(per_wave):.1f}")
for name, key in (("true cutoff", cut), ("detections", det.sum(1)), ("eta at truth", eta), ("detections then eta", det.sum(1) * 100 + eta)):
    order = np.argsort(-np.asarray(key, dtype=np.float64), kind="stable")
    s, ideal, nr = slowest(order, waves)
    print(f"sorted by {name:22s}: slowest wave {s:6.1f} (balanced ideal {ideal:.1f}, {nr} rounds)")

# ---- hybrid: per workgroup one QUAD wave (16 sites, four lanes each: n-loops a quarter as long) for the sites a static proxy ranks
# heaviest, three ordinary waves for the rest; sites dealt round-robin over the workgroups in proxy order
def hybrid(key, k=32, nq=16, fixed=8.0, fixed_q=12.0):
    order = np.argsort(-np.asarray(key, dtype=np.float64), kind="stable")
    worst = 0.0
    for g in range(k):
        mine = order[g::k]                       # this workgroup's sites, heaviest (by the proxy) first
        cq, cn = cut[mine[:nq]], cut[mine[nq:]]
        waves = [cq.max() / 4.0 + fixed_q] + [cn[i::3].max() + fixed for i in range(3)]
        worst = max(worst, max(waves))
    return worst

print("hybrid (k = 32, 16 quad sites per workgroup), slowest wave:")
for name, key in (("true cutoff (upper bound on what a proxy can do)", cut), ("detections", det.sum(1)),
                  ("detections, then more non-detections first", det.sum(1) * 100 - non.sum(1)),
                  ("log-odds proxy: detections / visits", det.sum(1) / np.maximum(det.sum(1) + non.sum(1), 1))):
    print(f"  proxy {name:52s}: {hybrid(key):5.1f}   (32 quad sites: {hybrid(key, nq=32):5.1f})")
