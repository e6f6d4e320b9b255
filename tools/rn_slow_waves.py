#!/usr/bin/env python3
"""occu_rn, BASELINE.json configs[3], with the waves' shares split by detections (rn_device.hpp): which of a chain's 128 main waves are the slow
ones?  Per-wave cycles from tools/stamps_rn_waves.py (argument) against what the waves hold at the posterior mean -- items, largest site, and the
sites whose FIRST floored non-detection lies within 20 nats of the best term (the evaluator's `rel`).   python tools/rn_slow_waves.py stamps.txt"""
import contextlib, io, os, sys
import numpy as np
from scipy.special import gammaln
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
STAMPS = sys.argv[1] if len(sys.argv) > 1 else 'profiles/r06/h_stamps_rn_waves_split.txt'
from biolith_amd.models import simulate_rn
with contextlib.redirect_stdout(io.StringIO()):
    data, truth = simulate_rn(n_sites=5000, n_site_covs=3, n_obs_covs=3, deployment_days_per_site=70, session_duration=7, random_seed=0)
Y = data["obs"][0, :, 0, :]; d = np.nansum(Y, axis=1).astype(int)
beta, alpha = np.array([0.116,-0.151,0.636,0.093]), np.array([-0.55,0.366,1.297,0.937])   # posterior means
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
rows=[]
for ln in open(STAMPS):
    p=ln.split()
    if len(p)==6 and p[0].isdigit(): rows.append([float(x) for x in p[1:]])
cyc=np.array(rows)
k, N = 32, 5000; nloc = -(-N // k)
items=np.zeros((k,4)); mx=np.zeros((k,4))
for b in range(k):
    s0, cnt = b*nloc, min(nloc, N-b*nloc)
    idx = np.arange(s0, s0+cnt); pos = idx[d[idx]>0]; npos=len(pos); spw=-(-npos//4)
    for w in range(4):
        blk = pos[w*spw:min(npos,(w+1)*spw)]
        items[b,w]=nch[blk].sum(); mx[b,w]=nch[blk].max()
c=cyc[:,:4]
print("corr cycles~items", np.corrcoef(c.ravel(), items.ravel())[0,1].round(2), " cycles~max nch", np.corrcoef(c.ravel(), mx.ravel())[0,1].round(2))
for lo,hi in ((0,44),(44,48),(48,52),(52,56),(56,70)):
    m=(items>=lo)&(items<hi)
    if m.sum(): print(f"items {lo}-{hi}: waves {m.sum():3d} mean cycles {c[m].mean():.0f} max {c[m].max():.0f}")
print("items per main wave: min/mean/max", items.min(), items.mean().round(1), items.max())
# features per main wave
rows=[]
for ln in open(STAMPS):
    p=ln.split()
    if len(p)==6 and p[0].isdigit(): rows.append([float(x) for x in p[1:]])
cyc=np.array(rows)[:32]
FL=-15.942385
nd=(Y==0)
lqnd=np.where(nd, lq, 0.0); cnon=lqnd.sum(1); lqmin=lqnd.min(1)
nstar=np.where(lqmin<0, FL/np.minimum(lqmin,-1e-30), 0.0)
Lmax=L.max(1)
rel=np.zeros(N,bool)
for i in range(N):
    if nstar[i]<=0: continue
    nn=np.arange(1, 8*nch[i]+1); nn=nn[nn>nstar[i]]
    if len(nn)==0: continue
    rel[i]=np.any(L[i,nn]+cnon[i]*(nstar[i]-nn) >= Lmax[i]-20.0)
print("rel sites", rel.sum(), "of", N)
feat={k:np.zeros((32,4)) for k in ("items","max","n3","n4","rel","sites")}
for b in range(k):
    s0, cnt = b*nloc, min(nloc, N-b*nloc)
    idx = np.arange(s0, s0+cnt); pos = idx[d[idx]>0]; npos=len(pos); spw=-(-npos//4)
    for w in range(4):
        blk = pos[w*spw:min(npos,(w+1)*spw)]
        feat["items"][b,w]=nch[blk].sum(); feat["max"][b,w]=nch[blk].max(); feat["n3"][b,w]=(nch[blk]>=3).sum(); feat["n4"][b,w]=(nch[blk]>=4).sum(); feat["rel"][b,w]=rel[blk].sum(); feat["sites"][b,w]=len(blk)
c=cyc[:,:4]
for kf,v in feat.items(): print(kf, "corr", np.corrcoef(c.ravel(), v.ravel())[0,1].round(2))
for r in range(0,4):
    m=feat["rel"]==r
    if m.sum(): print("rel sites in wave =",r,": waves",m.sum(),"mean cycles",c[m].mean().round(0))
slow=c>13500
print("slow waves:", slow.sum(), "with rel>0:", (slow&(feat['rel']>0)).sum(), "; fast waves with rel>0:", ((~slow)&(feat['rel']>0)).sum())
# second-floor relevance: after correcting the most negative non-detection exactly, is another visit's floor within 20 nats in the kept range?
srt=np.sort(lqnd,axis=1)   # ascending: most negative first
lq1=srt[:,0]; lq2=srt[:,1]; lq3=srt[:,2]
nstar2=np.where(lq2<0, FL/np.minimum(lq2,-1e-30), 0.0)
rel2=np.zeros(N,bool); rel3=np.zeros(N,bool)
for i in np.nonzero(d>0)[0]:
    nn=np.arange(0,101)
    corr1=np.maximum(0, FL-nn*lq1[i])
    Lc=L[i]+corr1
    kept=np.arange(1,8*nch[i]+1)
    if nstar2[i]>0:
        k2=kept[kept>nstar2[i]]
        if len(k2): rel2[i]=np.any(Lc[k2]+(cnon[i]-lq1[i])*(nstar2[i]-k2) >= Lc.max()-20)
print("d>0 sites:", (d>0).sum(), " rel (first floor)", (rel&(d>0)).sum(), " rel2 (second floor after the first is exact)", rel2.sum())
for m in sorted(set(feat["max"].ravel())):
    s = feat["max"] == m
    print(f"main waves whose largest site has {int(m)} items: {int(s.sum()):3d}, mean cycles {c[s].mean():.0f}, max {c[s].max():.0f}")
order = np.argsort(-c.ravel())[:14]
print("slowest main waves: cycles, items at the posterior mean, largest site:", [(int(c.ravel()[i]), int(feat["items"].ravel()[i]), int(feat["max"].ravel()[i])) for i in order])
for lo, hi in ((0, 40), (40, 46), (46, 52), (52, 70)):
    m = (feat["items"] >= lo) & (feat["items"] < hi)
    print(f"items {lo}-{hi}: waves {int(m.sum()):3d} mean cycles {c[m].mean():.0f}")
