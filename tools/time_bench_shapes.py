#!/usr/bin/env python3
"""us per leapfrog (slowest chain) of the bench workloads' shapes, for A/B variant libraries:
    python tools/time_bench_shapes.py lib.so [lib.so ...]  [-- workload ...]      (each library in its own process)"""
import contextlib, io, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
if len(sys.argv) > 1 and sys.argv[1] == "child":
    import numpy as np
    import bench
    from biolith_amd.engine import OccuDataset
    from biolith_amd.models import simulate, simulate_dyn, simulate_rn
    for name in sys.argv[2:]:
        wl = bench.WORKLOADS[name]
        with contextlib.redirect_stdout(io.StringIO()):
            d, _ = {"occu_rn": simulate_rn, "occu_dyn": simulate_dyn}.get(wl["model"], simulate)(**wl["cfg"])
        ds = OccuDataset(d["site_covs"], d["obs_covs"], d["obs"], model=wl["model"], **wl.get("options", {}))
        C = bench.workload_chains(wl)
        us = []
        for s in range(4):
            r = ds.nuts(num_warmup=wl["num_warmup"], num_samples=wl["num_samples"], num_chains=C, seed=s)
            us.append(1e3 * r.kernel_ms / r.n_leapfrog.reshape(C, -1).sum(axis=1).max())
        print(f"  {name:14s} {np.mean(us[1:]):7.3f} us/leapfrog of the slowest chain (seeds 1-3: {' '.join(f'{u:.3f}' for u in us[1:])})  k={r.wgs_per_chain} {r.kernel_name.strip()}", flush=True)
        ds.close()
else:
    args = sys.argv[1:]
    wls = ["occu", "occu_stacked", "occu_dyn"]
    if "--" in args:
        i = args.index("--"); wls = args[i + 1:]; args = args[:i]
    for lib in args or [os.path.join("biolith_amd", "lib", "libbiolith_hip.so")]:
        print(lib, flush=True)
        subprocess.run([sys.executable, os.path.abspath(__file__), "child"] + wls, env=dict(os.environ, BIOLITH_HIP_LIB=os.path.join(ROOT, lib)))
