#!/usr/bin/env python3
"""occu_rn at BASELINE.json's config 4 (5 000 sites x 10 visits, 3 + 3 covariates, max_abundance 100, 4 chains):
K1 parity against the float64 oracle at full size, the oracle's first trees, and the time per leapfrog.
    python tools/time_rn.py [lib.so ...]        (each library in its own process; default: the shipped one)"""
import contextlib, io, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
if len(sys.argv) > 1 and sys.argv[1] == "child":
    import numpy as np
    import oracle
    from biolith_amd.engine import OccuDataset
    from biolith_amd.models import simulate_rn
    with contextlib.redirect_stdout(io.StringIO()):
        d, truth = simulate_rn(n_sites=5000, n_site_covs=3, n_obs_covs=3, deployment_days_per_site=70, session_duration=7)
    od = oracle.OracleData(d["site_covs"], d["obs_covs"], d["obs"], model="occu_rn")
    ds = OccuDataset(d["site_covs"], d["obs_covs"], d["obs"], model="occu_rn")
    th0 = np.concatenate([truth["beta"][0], truth["alpha"][0]])
    th = np.concatenate([th0[None] + np.random.default_rng(0).normal(0, 0.1, size=(3, 8)), np.random.default_rng(1).uniform(-2, 2, size=(3, 8))])
    th = th.astype(np.float32).astype(np.float64)
    Uo, Go = od.potential_grad(th)
    Ug, Gg = ds.logp_grad(th)
    print("  K1 rel dU", np.max(np.abs(Ug - Uo) / np.abs(Uo)), " max|dG|/max|G| per theta", (np.abs(Gg - Go).max(1) / np.abs(Go).max(1)).round(7).tolist())
    o = oracle.nuts_run(od, 0, 3, num_chains=2, seed=3)
    r = ds.nuts(num_warmup=0, num_samples=3, num_chains=2, seed=3)
    print("  first trees", o["num_steps"].tolist(), r.num_steps.tolist(), "max |d draw|", float(np.abs(o["draws"] - r.draws).max()))
    for s in range(3):
        r = ds.nuts(num_warmup=1000, num_samples=1000, num_chains=4, seed=s)
        per_chain = r.n_leapfrog.sum(axis=1)
        print(f"  cfg4 seed {s}: kernel {r.kernel_ms:8.2f} ms  {1e3 * r.kernel_ms / per_chain.max():6.2f} us/leapfrog (slowest chain {int(per_chain.max())} of {int(per_chain.sum())}; {1e3 * r.kernel_ms / per_chain.mean():6.2f} by the chains' mean, the bench line's figure)"
              f"  k={r.wgs_per_chain} div {int(r.diverging.sum())} l2local {r.chains_l2_local} means {r.draws.reshape(-1, 8).mean(0).round(3).tolist()}")
else:
    libs = sys.argv[1:] or [os.path.join("biolith_amd", "lib", "libbiolith_hip.so")]
    for lib in libs:
        print(lib, flush=True)
        subprocess.run([sys.executable, os.path.abspath(__file__), "child"], env=dict(os.environ, BIOLITH_HIP_LIB=os.path.join(ROOT, lib)))
