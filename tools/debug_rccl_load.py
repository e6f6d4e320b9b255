#!/usr/bin/env python3
"""Which ROCm stack does each piece bind to?  usage: python tools/debug_rccl_load.py [torch-first|lib-first|no-torch]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
order = sys.argv[1] if len(sys.argv) > 1 else "lib-first"
if order == "torch-first":
    import torch
from biolith_amd import _ffi
_ffi.load()
if order == "lib-first":
    import torch
import numpy as np
from biolith_amd.engine import OccuDataset
from biolith_amd.distributed import comm_from_env, gather_draws, rccl_version
rng = np.random.default_rng(0)
X = rng.normal(size=(200, 2)); W = rng.normal(size=(200, 1, 4, 2)); Y = (rng.uniform(size=(1, 200, 1, 4)) < 0.3) * 1.0
ds = OccuDataset(X, W, Y)
ds.launch(num_warmup=20, num_samples=10, num_chains=2, seed=1)
while not ds.done():
    pass
ds.wait()
print("rccl", rccl_version(), flush=True)
def maps():
    seen = []
    for ln in open("/proc/self/maps"):
        p = ln.split()[-1]
        if any(k in p for k in ("libamdhip64", "librccl", "libhsa-runtime", "librocm_smi")) and p not in seen:
            seen.append(p)
    return seen
print("\n".join(maps()), flush=True)
try:
    comm = comm_from_env(0, rank=0, world=1)
    r = gather_draws([comm], [ds], [2])
    print("OK", order, r.draws.shape, np.array_equal(r.draws, ds.fetch().draws))
except Exception as e:
    print("FAIL", order, e)
