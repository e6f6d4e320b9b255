#!/usr/bin/env python3
"""Per-leapfrog time of the secondary models on comparable synthetic sizes (4 chains x (300 + 300))."""
import contextlib, io, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from biolith_amd.engine import OccuDataset
from biolith_amd.models import simulate, simulate_cop, simulate_nmixture, simulate_rn

def run(name, ds):
    r = ds.nuts(num_warmup=300, num_samples=300, num_chains=4, seed=0)
    r = ds.nuts(num_warmup=300, num_samples=300, num_chains=4, seed=1)
    # the launch ends with its slowest chain: time per leapfrog of THAT chain (a mean over chains would charge the others' idle
    # tail to every leapfrog -- occu_cop's false-positive posterior is bimodal and its chains' tree sizes differ a lot)
    per_chain = r.n_leapfrog.sum(axis=1)
    print(f"{name:44s} kernel {r.kernel_ms:8.2f} ms  {1e3 * r.kernel_ms / per_chain.max():7.2f} us/leapfrog (slowest chain: {int(per_chain.max())} of "
          f"{int(per_chain.sum())} leapfrogs)  k={r.wgs_per_chain} D={ds.D}")

with contextlib.redirect_stdout(io.StringIO()):
    kw = dict(n_sites=5000, n_site_covs=3, n_obs_covs=3, deployment_days_per_site=70, session_duration=7)
    d0, _ = simulate(**kw)
    dfp, _ = simulate(**kw, prob_fp_constant=0.1)
    drn, _ = simulate_rn(**kw)
    dcop, _ = simulate_cop(**kw)
    dnm, _ = simulate_nmixture(**kw, min_abundance=0.5, max_abundance=8.0, max_observation_rate=6.0)
    dst, _ = simulate(n_sites=2000, n_periods=8, n_site_covs=3, n_obs_covs=3, deployment_days_per_site=28, session_duration=7)
run("occu 5000x10", OccuDataset(d0["site_covs"], d0["obs_covs"], d0["obs"]))
run("occu stacked 2000x8x4 (configs[4] stand-in)", OccuDataset(dst["site_covs"], dst["obs_covs"], dst["obs"]))
run("occu_fp constant 5000x10", OccuDataset(dfp["site_covs"], dfp["obs_covs"], dfp["obs"], model="occu_fp", fp_mode="constant"))
run("occu_cop (fp constant) 5000x10", OccuDataset(dcop["site_covs"], dcop["obs_covs"], dcop["obs"], model="occu_cop", fp_mode="constant",
                                                   session_duration=dcop["session_duration"]))
K = int(np.nanmax(dnm["obs"])) + 10
run(f"nmixture K={K} 5000x10", OccuDataset(dnm["site_covs"], dnm["obs_covs"], dnm["obs"], model="nmixture", max_abundance=K))
run("nmixture K=100 5000x10", OccuDataset(dnm["site_covs"], dnm["obs_covs"], dnm["obs"], model="nmixture", max_abundance=100))
run("occu_rn K=100 5000x10", OccuDataset(drn["site_covs"], drn["obs_covs"], drn["obs"], model="occu_rn"))
# the generators' own default sizes (100 sites x 52 visits: what the reference's tests fit), lanes sharing a site pair / one pair per lane
import os
with contextlib.redirect_stdout(io.StringIO()):
    c1, _ = simulate_cop()
    n1, _ = simulate_nmixture()
for g in ("", "1"):
    if g:
        os.environ["BIOLITH_HIP_OCCU_G"] = g
    else:
        os.environ.pop("BIOLITH_HIP_OCCU_G", None)
    tag = "host's choice" if not g else "one pair per lane"
    run(f"occu_cop defaults 100x52 ({tag})", OccuDataset(c1["site_covs"], c1["obs_covs"], c1["obs"], model="occu_cop", fp_mode=None, session_duration=c1["session_duration"]))
    run(f"nmixture defaults 100x52 ({tag})", OccuDataset(n1["site_covs"], n1["obs_covs"], n1["obs"], model="nmixture", max_abundance=int(np.nanmax(n1["obs"])) + 10))

