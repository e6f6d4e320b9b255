"""Configurations VERDICT r01 listed as never run on hardware: the stacked-period stand-in of BASELINE.json configs[4] at its
stated size, and launches of more than 8 chains per device (the `per_xcd = 2` branch of choose_geometry)."""
import json
import os

import numpy as np
import pytest

import oracle
from biolith_amd.engine import OccuDataset
from biolith_amd.evaluation import effective_sample_size, split_gelman_rubin
from conftest import GOLDEN, quiet_simulate

pytestmark = pytest.mark.gpu

CFG5 = dict(n_sites=2000, n_periods=8, n_site_covs=3, n_obs_covs=3, deployment_days_per_site=28, session_duration=7)


def test_config5_standin_at_size_no_reference_counterpart():
    """BASELINE.json configs[4] names a dynamic (colonisation / extinction) model that the reference does NOT have (SURVEY.md
    section 0.7): NO REFERENCE COUNTERPART.  The nearest reference behaviour is stacked periods sharing psi (occu.py:198-210),
    run here at the stated size 2000 sites x 8 periods x 4 visits: K1 parity, first trees, and 4 x (500 + 2000) against the
    oracle's captured posterior (tests/golden/oracle_posterior_cfg5.json, make_oracle_posterior.py cfg5: 4 x (1000 + 1000)) at
    SURVEY.md section 8c's tolerances -- mean within 4 MCSE, sd within 10 %, split R-hat < 1.01 -- as everywhere else.
    (Round 4: the kernel shares a site pair among 8 lanes here -- one period each -- occu_device.hpp: bl_eval_sites_grp.)"""
    data, truth, _ = quiet_simulate(**CFG5)
    assert data["obs"].shape == (1, 2000, 8, 4)
    od = oracle.OracleData(data["site_covs"], data["obs_covs"], data["obs"])
    ds = OccuDataset(data["site_covs"], data["obs_covs"], data["obs"])
    th = np.random.default_rng(5).uniform(-2, 2, size=(4, od.D)).astype(np.float32).astype(np.float64)
    U0, G0 = od.potential_grad(th)
    U1, G1 = ds.logp_grad(th)
    assert np.max(np.abs(U1 - U0) / np.abs(U0)) < 1e-6
    assert np.max(np.abs(G1 - G0)) < 1e-5 * np.max(np.abs(G0))
    o = oracle.nuts_run(od, 0, 4, num_chains=2, seed=3)
    r = ds.nuts(num_warmup=0, num_samples=4, num_chains=2, seed=3)
    assert np.array_equal(o["num_steps"][:, :3], r.num_steps[:, :3]), (o["num_steps"], r.num_steps)
    assert np.allclose(o["draws"][:, 0], r.draws[:, 0], atol=2e-3)
    fx = json.load(open(os.path.join(GOLDEN, "oracle_posterior_cfg5.json")))
    r = ds.nuts(num_warmup=500, num_samples=2000, num_chains=4, seed=0)
    assert r.lds_staged and r.diverging.sum() == 0
    assert r.lane_group[0] * r.lane_group[1] > 1, r.lane_group   # 32 visits per site: lanes share a pair
    flat = r.draws.reshape(-1, od.D).astype(np.float64)
    mcse = np.sqrt(flat.var(0) / effective_sample_size(r.draws) + np.array(fx["sd"]) ** 2 / np.array(fx["ess"]))
    assert np.all(np.abs(flat.mean(0) - fx["mean"]) <= 4 * mcse), (flat.mean(0) - fx["mean"], mcse)
    assert np.all(np.abs(flat.std(0) / fx["sd"] - 1) < 0.1)
    assert np.all(np.abs(flat.mean(0) - fx["map"]) < 3 * np.array(fx["laplace_sd"]))
    assert split_gelman_rubin(r.draws).max() < 1.01
    assert abs(r.num_steps.mean() / fx["mean_num_steps"] - 1) < 0.15
    psi, _ = ds.deterministic(flat[::10])
    assert abs(psi.mean() - truth["z"].mean()) < 0.1


@pytest.mark.parametrize("chains", [9, 16])
def test_more_than_eight_chains_per_device_build_the_oracles_trees(chains):
    """9 and 16 chains: two chains share an XCD, so a chain has at most 16 workgroups (choose_geometry's per_xcd = 2)."""
    data, _, _ = quiet_simulate(n_sites=3000, n_site_covs=2, n_obs_covs=2, deployment_days_per_site=42, session_duration=7, random_seed=2)
    od = oracle.OracleData(data["site_covs"], data["obs_covs"], data["obs"])
    ds = OccuDataset(data["site_covs"], data["obs_covs"], data["obs"])
    o = oracle.nuts_run(od, 10, 6, num_chains=chains, seed=11)
    r = ds.nuts(num_warmup=10, num_samples=6, num_chains=chains, seed=11)
    assert r.wgs_per_chain <= 16 and r.lds_staged
    assert np.array_equal(o["num_steps"][:, :3], r.num_steps[:, :3]), (o["num_steps"], r.num_steps)
    assert (o["num_steps"] == r.num_steps).mean() >= 0.8
    assert np.allclose(o["draws"][:, 0], r.draws[:, 0], atol=2e-2)
    # chains do not depend on how many run together: chain 8 of this launch = chain 0 of a launch with chain_offset 8
    one = ds.nuts(num_warmup=10, num_samples=6, num_chains=1, seed=11, chain_offset=8)
    assert np.array_equal(one.draws[0], r.draws[8])
    long = ds.nuts(num_warmup=200, num_samples=200, num_chains=chains, seed=1)
    assert long.diverging.sum() == 0 and split_gelman_rubin(long.draws).max() < 1.05
