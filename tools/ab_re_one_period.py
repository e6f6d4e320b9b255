#!/usr/bin/env python3
"""A/B of the random-effects kernel's compile-time-fact forms (re_kernel.hpp: EFF) through BIOLITH_HIP_RE_EFF (0: general kernel, 1: effects
as facts without the one-period fact, unset: the host's choice): first that the forms give the same draws bit for bit, then the time per
leapfrog on four shapes at capacity (3, 3) -- so that it also runs on a (3,3)-only variant library:
    make -C biolith_amd/csrc variant NAME=t1all EXTRA=-DBL_RE_T1_ALL    (round 4: the one-period fact for the forms with observation effects)
    BIOLITH_HIP_LIB=$PWD/biolith_amd/lib/libbiolith_hip_t1all.so python tools/ab_re_one_period.py
Identical draws => identical trajectories: the figures compare kernels, not how evenly four chains happened to adapt."""
import contextlib, io, os, sys, numpy as np
sys.path.insert(0, os.getcwd())
from biolith_amd.engine import OccuDataset
from biolith_amd.models import simulate
rng = np.random.default_rng(17)
N, J = 150, 7
X = rng.normal(size=(N, 3)).astype(np.float32); W = rng.normal(size=(N, 1, J, 3)).astype(np.float32)
Y = (rng.uniform(size=(1, N, 1, J)) < 0.4).astype(np.float32)
for site, obs in ((False, True), (True, True)):
    ds = OccuDataset(X, W, Y, model="occu_re", site_random_effects=site, obs_random_effects=obs)
    init = rng.uniform(-0.5, 0.5, size=(2, ds.D))
    out = {}
    for knob in ("0", "1", ""):
        if knob: os.environ["BIOLITH_HIP_RE_EFF"] = knob
        else: os.environ.pop("BIOLITH_HIP_RE_EFF", None)
        r = ds.nuts(num_warmup=40, num_samples=20, num_chains=2, seed=4, init_theta=init, wgs_per_chain=2)
        out[knob] = r
        print(site, obs, repr(knob), r.kernel_name, "same draws as general:", np.array_equal(r.draws, out["0"].draws), np.array_equal(r.num_steps, out["0"].num_steps), flush=True)
    ds.close()
with contextlib.redirect_stdout(io.StringIO()):
    d_small, _ = simulate(n_site_covs=3, n_obs_covs=3, simulate_missing=True)
    d_big, _ = simulate(n_sites=2000, n_site_covs=3, n_obs_covs=3, deployment_days_per_site=70, session_duration=7, site_random_effects=True)
for name, d, kw in (("obs 100x52", d_small, dict(obs_random_effects=True)), ("both 100x52", d_small, dict(site_random_effects=True, obs_random_effects=True)),
                    ("obs 2000x10", d_big, dict(obs_random_effects=True)), ("both 2000x10", d_big, dict(site_random_effects=True, obs_random_effects=True))):
    ds = OccuDataset(d["site_covs"], d["obs_covs"], d["obs"], model="occu_re", **kw)
    for p in (1, 2):
        for knob in ("1", ""):
            if knob: os.environ["BIOLITH_HIP_RE_EFF"] = knob
            else: os.environ.pop("BIOLITH_HIP_RE_EFF", None)
            ds.nuts(num_warmup=300, num_samples=300, num_chains=4, seed=0)
            r = ds.nuts(num_warmup=300, num_samples=300, num_chains=4, seed=1)
            per = r.n_leapfrog.reshape(4, -1).sum(axis=1)
            print(f"{name:14s} pass {p} knob {knob!r:4s} {r.kernel_name} kernel {r.kernel_ms:8.2f} ms  {1e3 * r.kernel_ms / per.mean():6.3f} us/leapfrog (mean)  {1e3 * r.kernel_ms / per.max():6.3f} (slowest chain)", flush=True)
    ds.close()
