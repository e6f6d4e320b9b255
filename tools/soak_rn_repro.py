#!/usr/bin/env python3
"""Replays tools/soak_rn.py's random stream up to its n-th launch and reports K1 against the oracle there, per theta (diagnosis)."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import oracle
from biolith_amd.engine import OccuDataset
target = (9000, 14, 1, 0.0)
rng = np.random.default_rng(1)
n = 0
while True:
    N = int(rng.choice([37, 160, 700, 2500, 5000, 9000])); J = int(rng.choice([3, 7, 10, 14])); T = int(rng.choice([1, 1, 1, 2, 3]))
    share = float(rng.choice([0.0, 0.1, 0.34, 0.6, 0.95, 1.0]))
    X = rng.normal(size=(N, 2)).astype(np.float32); W = rng.normal(size=(N, T, J, 2)).astype(np.float32)
    Y = (rng.uniform(size=(1, N, T, J)) < rng.uniform(0.1, 0.6)) * 1.0
    Y[0, rng.uniform(size=N) < share] = 0.0
    Y[rng.uniform(size=Y.shape) < 0.1] = np.nan
    C = int(rng.choice([1, 2, 4])); nw = int(rng.choice([30, 150, 400])); nsamp = int(rng.choice([30, 200])); seed = int(rng.integers(1 << 30))
    if (N, J, T, share) == target and n % 10 == 0:
        break
    n += 1
    if n > 400: raise SystemExit("not found")
print("launch", n, "C", C, "warmup", nw, "samples", nsamp, "seed", seed, "detections share", np.nanmean(Y))
np.savez("gpurun_out/soak_case.npz", X=X, W=W, Y=Y)
ds = OccuDataset(X, W, Y.astype(np.float32), model="occu_rn")
od = oracle.OracleData(X, W, Y, model="occu_rn")
if os.path.exists("gpurun_out/soak_theta.npy") and os.environ.get("REPRO_REUSE") == "1":
    th = np.load("gpurun_out/soak_theta.npy")
else:
    r = ds.nuts(num_warmup=nw, num_samples=nsamp, num_chains=C, seed=seed)
    th = r.draws[:, -1].astype(np.float64)
    np.save("gpurun_out/soak_theta.npy", th)
    print("steps", r.num_steps.mean(), "step size", r.step_size)
print("theta", th.round(3).tolist())
Uo, Go = od.potential_grad(th); Ug, Gg = ds.logp_grad(th)
print("rel dU", np.abs(Ug - Uo) / np.abs(Uo), "U", Uo)
print("rel dG", (np.abs(Gg - Go) / np.abs(Go).max(1, keepdims=True)).round(7).tolist())
print("G oracle", Go.round(3).tolist()); print("G engine", Gg.round(3).tolist())
th2 = np.concatenate([th, np.random.default_rng(0).uniform(-1, 1, size=(3, th.shape[1]))])
Uo, Go = od.potential_grad(th2); Ug, Gg = ds.logp_grad(th2)
print("more thetas: rel dU", (np.abs(Ug - Uo) / np.abs(Uo)).tolist(), "rel dG", (np.abs(Gg - Go).max(1) / np.abs(Go).max(1)).tolist())
