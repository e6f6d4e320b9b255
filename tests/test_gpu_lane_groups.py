"""Lanes that share one site pair (occu_device.hpp: bl_eval_sites_grp; SURVEY.md section 7.1 "lanes-over-visits"): the plain occupancy
model and its false-positive form at many visits per site -- simulate()'s own defaults (100 sites x 52 visits, occu.py:251-252, 336),
the reference's benchmark grid (benchmarks/occu_spoccupancy.py:16-70) and stacked periods (occu.py:198-210).

Every group size G = period lanes x visit lanes must give the one-pair-per-lane kernel's results up to the order of float32 sums:
K1 against the float64 oracle at the usual 1e-6 / 1e-5, and -- both sides on the same xoshiro streams -- the oracle's first trees."""
import os

import numpy as np
import pytest

import oracle
from biolith_amd.engine import OccuDataset
from conftest import load_golden, quiet_simulate

pytestmark = pytest.mark.gpu
U_RTOL, G_RTOL = 1e-6, 1e-5


@pytest.fixture(params=["single", "multi"])
def force_group(monkeypatch, request):
    """Forces the lanes per pair; once with the small problems' one-workgroup form allowed (k = 1, 7 compute waves, no exchange --
    taken when the shape qualifies), once with it switched off (the multi-workgroup form at the same group size)."""
    monkeypatch.setenv("BIOLITH_HIP_SINGLE", "1" if request.param == "single" else "0")

    def force(g, gt=None):
        monkeypatch.setenv("BIOLITH_HIP_OCCU_G", str(g))
        if gt is None:
            monkeypatch.delenv("BIOLITH_HIP_OCCU_GT", raising=False)
        else:
            monkeypatch.setenv("BIOLITH_HIP_OCCU_GT", str(gt))
    yield force


def _stacked(n_sites=300, **kw):
    d, _, _ = quiet_simulate(n_sites=n_sites, n_periods=4, n_site_covs=2, n_obs_covs=3, deployment_days_per_site=42, session_duration=7,
                             simulate_missing=True, random_seed=3, **kw)
    return d


# (group size, log2 of the period lanes or None = the host's split)
GROUPS = [(1, None), (2, None), (4, None), (8, None), (16, None), (2, 0), (4, 0), (4, 1), (8, 1), (16, 2)]


@pytest.mark.parametrize("g,gt", GROUPS)
@pytest.mark.parametrize("name", ["default", "missing", "missing_3periods", "stacked"])
def test_k1_parity_for_every_group_size(name, g, gt, force_group):
    d = _stacked() if name == "stacked" else load_golden(name)
    od = oracle.OracleData(d["site_covs"], d["obs_covs"], d["obs"])
    ds = OccuDataset(d["site_covs"], d["obs_covs"], d["obs"])
    th = np.random.default_rng(4).uniform(-2, 2, size=(4, od.D)).astype(np.float32).astype(np.float64)
    Uo, Go = od.potential_grad(th)
    force_group(g, gt)
    Ug, Gg = ds.logp_grad(th)
    assert np.max(np.abs(Ug - Uo) / np.abs(Uo)) <= U_RTOL, (Ug, Uo)
    assert np.max(np.abs(Gg - Go)) <= G_RTOL * np.max(np.abs(Go)), np.max(np.abs(Gg - Go))
    ds.close()


@pytest.mark.parametrize("n_sites", [1, 2, 3, 47, 48, 49, 97, 385])
@pytest.mark.parametrize("g", [2, 8, 16])
def test_ragged_slices_with_groups(n_sites, g, force_group):
    """Odd site counts (a dummy second site in the last pair), slices that leave lane groups without a pair, one workgroup or several."""
    rng = np.random.default_rng(n_sites)
    T, J = 3, 7
    X = rng.normal(size=(n_sites, 2)); W = rng.normal(size=(n_sites, T, J, 2))
    Y = (rng.uniform(size=(1, n_sites, T, J)) < 0.3) * 1.0
    Y[rng.uniform(size=Y.shape) < 0.15] = np.nan
    Y[0, 0, 1, :] = np.nan  # a period without a single visit
    od = oracle.OracleData(X, W, Y)
    ds = OccuDataset(X, W, Y)
    th = rng.uniform(-1.5, 1.5, size=(2, 6)).astype(np.float32).astype(np.float64)
    Uo, Go = od.potential_grad(th)
    force_group(g)
    Ug, Gg = ds.logp_grad(th)
    assert np.max(np.abs(Ug - Uo) / np.abs(Uo)) <= U_RTOL
    assert np.max(np.abs(Gg - Go)) <= G_RTOL * np.max(np.abs(Go))
    ds.close()


@pytest.mark.parametrize("ks,ko", [(0, 0), (1, 4), (5, 2), (16, 16)])
def test_groups_at_other_covariate_capacities(ks, ko, force_group):
    rng = np.random.default_rng(100 * ks + ko)
    N, T, J = 150, 2, 9
    X = rng.normal(size=(N, ks)) * 0.5; W = rng.normal(size=(N, T, J, ko)) * 0.5
    Y = (rng.uniform(size=(1, N, T, J)) < 0.3) * 1.0
    od = oracle.OracleData(X, W, Y)
    ds = OccuDataset(X, W, Y)
    th = (rng.uniform(-1, 1, size=(2, ks + ko + 2)) * 0.7).astype(np.float32).astype(np.float64)
    Uo, Go = od.potential_grad(th)
    for g in (4, 16):
        force_group(g)
        Ug, Gg = ds.logp_grad(th)
        assert np.max(np.abs(Ug - Uo) / np.abs(Uo)) <= U_RTOL
        assert np.max(np.abs(Gg - Go)) <= G_RTOL * np.max(np.abs(Go))
    ds.close()


@pytest.mark.parametrize("g,gt", [(1, None), (2, None), (4, None), (8, None), (16, None), (4, 0), (8, 1)])
@pytest.mark.parametrize("name,seed", [("default", 1), ("missing_3periods", 2)])
def test_first_trees_are_the_oracles_for_every_group_size(name, seed, g, gt, force_group):
    """Same xoshiro streams on both sides: the first transitions build the oracle's trees whatever the lanes per pair."""
    d = load_golden(name)
    od = oracle.OracleData(d["site_covs"], d["obs_covs"], d["obs"])
    ds = OccuDataset(d["site_covs"], d["obs_covs"], d["obs"])
    force_group(g, gt)
    want = (1 << (gt if gt is not None else 0))
    # the very first transitions (no adaptation in front of them): the same trees, the same positions
    o = oracle.nuts_run(od, 0, 4, num_chains=3, seed=seed)
    r = ds.nuts(num_warmup=0, num_samples=4, num_chains=3, seed=seed)
    assert r.lane_group[0] * r.lane_group[1] == g and (gt is None or r.lane_group[0] == want), r.lane_group
    assert np.array_equal(o["num_steps"][:, :2], r.num_steps[:, :2]), (o["num_steps"], r.num_steps)
    assert np.allclose(o["draws"][:, 0], r.draws[:, 0], atol=2e-3)
    # ... and through the first adaptation steps: float32-vs-float64 rounding is amplified by the dynamics by then (a U-turn test that
    # is decided in the last bits may flip), so: most trees equal, positions and step sizes close
    W, S = 12, 8
    o = oracle.nuts_run(od, W, S, num_chains=3, seed=seed)
    r = ds.nuts(num_warmup=W, num_samples=S, num_chains=3, seed=seed)
    assert (o["num_steps"] == r.num_steps).mean() >= 0.8, (o["num_steps"], r.num_steps)
    assert np.allclose(o["draws"][:, 0], r.draws[:, 0], atol=2e-2)
    assert np.allclose(o["step_size"], r.step_size, rtol=0.05)
    ds.close()


@pytest.mark.parametrize("g", [1, 4, 16])
def test_false_positive_model_with_groups(g, force_group):
    d, _, _ = quiet_simulate(n_sites=200, n_periods=2, n_site_covs=2, n_obs_covs=2, deployment_days_per_site=84, session_duration=7,
                             prob_fp_constant=0.1, simulate_missing=True, random_seed=5)
    for mode in ("constant", "unoccupied"):
        od = oracle.OracleData(d["site_covs"], d["obs_covs"], d["obs"], model="occu_fp", fp_mode=mode)
        ds = OccuDataset(d["site_covs"], d["obs_covs"], d["obs"], model="occu_fp", fp_mode=mode)
        th = np.random.default_rng(6).uniform(-1.5, 1.5, size=(3, od.D)).astype(np.float32).astype(np.float64)
        Uo, Go = od.potential_grad(th)
        force_group(g)
        Ug, Gg = ds.logp_grad(th)
        assert np.max(np.abs(Ug - Uo) / np.abs(Uo)) <= U_RTOL, (mode, Ug, Uo)
        assert np.max(np.abs(Gg - Go)) <= G_RTOL * np.max(np.abs(Go)), (mode, np.max(np.abs(Gg - Go)))
        init = np.tile(np.concatenate([np.zeros(od.D - 1), [-2.0]]), (2, 1))
        o = oracle.nuts_run(od, 8, 6, num_chains=2, seed=3, init=init)
        r = ds.nuts(num_warmup=8, num_samples=6, num_chains=2, seed=3, init_theta=init)
        assert np.array_equal(o["num_steps"][:, :3], r.num_steps[:, :3]), (mode, o["num_steps"], r.num_steps)
        ds.close()


def test_the_hosts_choice():
    """One pair per lane while a pair has few visits (the headline: 5); groups at simulate()'s defaults and with stacked periods."""
    for var in ("BIOLITH_HIP_OCCU_G", "BIOLITH_HIP_OCCU_GT", "BIOLITH_HIP_GRP_VISITS"):
        assert var not in os.environ
    d = load_golden("default")                      # 100 sites x 52 visits
    ds = OccuDataset(d["site_covs"], d["obs_covs"], d["obs"])
    r = ds.nuts(num_warmup=5, num_samples=5, num_chains=2, seed=0)
    # a small problem: the whole chain on ONE workgroup of 7 compute waves (no exchange), 8 lanes per pair
    assert r.lane_group == (1, 8) and r.wgs_per_chain == 1 and r.threads_per_wg == 512, (r.lane_group, r.wgs_per_chain, r.threads_per_wg)
    os.environ["BIOLITH_HIP_SINGLE"] = "0"          # ... and with that form switched off: several workgroups, 16 lanes per pair
    try:
        r = ds.nuts(num_warmup=5, num_samples=5, num_chains=2, seed=0)
    finally:
        del os.environ["BIOLITH_HIP_SINGLE"]
    assert r.lane_group == (1, 16) and r.wgs_per_chain > 1 and r.threads_per_wg == 256, (r.lane_group, r.wgs_per_chain)
    ds.close()
    d = _stacked(2000)                              # 4 periods x 6 visits
    ds = OccuDataset(d["site_covs"], d["obs_covs"], d["obs"])
    r = ds.nuts(num_warmup=5, num_samples=5, num_chains=4, seed=0)
    assert r.lane_group == (4, 1), r.lane_group     # the periods first: those lanes exchange nothing
    ds.close()
    d, _, _ = quiet_simulate(n_sites=3000, n_site_covs=3, n_obs_covs=3, deployment_days_per_site=35, session_duration=7)
    ds = OccuDataset(d["site_covs"], d["obs_covs"], d["obs"])
    r = ds.nuts(num_warmup=5, num_samples=5, num_chains=4, seed=0)
    assert r.lane_group == (1, 1) and r.threads_per_wg == 256, (r.lane_group, r.threads_per_wg)
    ds.close()


def test_posterior_with_groups_matches_the_one_pair_per_lane_kernel(force_group):
    """Distributional check on simulate()'s defaults: 4 x (300 + 1000) draws either way."""
    from biolith_amd.evaluation import effective_sample_size, split_gelman_rubin

    d = load_golden("default")
    ds = OccuDataset(d["site_covs"], d["obs_covs"], d["obs"])
    force_group(1)
    a = ds.nuts(num_warmup=300, num_samples=1000, num_chains=4, seed=0)
    force_group(16)
    b = ds.nuts(num_warmup=300, num_samples=1000, num_chains=4, seed=50)
    fa, fb = a.draws.reshape(-1, ds.D).astype(np.float64), b.draws.reshape(-1, ds.D).astype(np.float64)
    mcse = np.sqrt(fa.var(0) / effective_sample_size(a.draws) + fb.var(0) / effective_sample_size(b.draws))
    assert np.all(np.abs(fa.mean(0) - fb.mean(0)) <= 4 * mcse)
    assert np.all(np.abs(fa.std(0) / fb.std(0) - 1) < 0.1)
    assert split_gelman_rubin(b.draws).max() < 1.01
    ds.close()


@pytest.mark.parametrize("i", [4, 6, 7])
def test_benchmark_grid_rows_at_full_size(i):
    """Rows of the reference's own benchmark grid (benchmarks/occu_spoccupancy.py:16-70: 100 * 2^i sites x int(8 * 2^(i/2)) visits, 2 + 1
    covariates, one chain) at full size with the host's own choice of lanes per pair: K1 against the oracle and its first trees."""
    n, j = 100 * 2 ** i, int(8 * 2 ** (i / 2))
    d, _, _ = quiet_simulate(n_site_covs=2, n_obs_covs=1, n_sites=n, deployment_days_per_site=7 * j, session_duration=7, random_seed=42 + i)
    assert d["obs"].shape == (1, n, 1, j)
    od = oracle.OracleData(d["site_covs"], d["obs_covs"], d["obs"])
    ds = OccuDataset(d["site_covs"], d["obs_covs"], d["obs"])
    th = np.random.default_rng(i).uniform(-2, 2, size=(3, od.D)).astype(np.float32).astype(np.float64)
    Uo, Go = od.potential_grad(th)
    Ug, Gg = ds.logp_grad(th)
    assert np.max(np.abs(Ug - Uo) / np.abs(Uo)) <= U_RTOL
    assert np.max(np.abs(Gg - Go)) <= G_RTOL * np.max(np.abs(Go))
    o = oracle.nuts_run(od, 0, 3, num_chains=1, seed=i)
    r = ds.nuts(num_warmup=0, num_samples=3, num_chains=1, seed=i)
    assert r.lds_staged and r.lane_group[0] * r.lane_group[1] >= 2, r.lane_group      # 32 / 64 / 90 visits per site: lanes share a pair
    if i == 7:   # 12 800 x 90 = 9.2 MB of records: more than one XCD's LDS -- the wide geometry (a chain across XCDs) WITH lane groups (round 5)
        assert r.wgs_per_chain > 32 and r.lane_group[0] * r.lane_group[1] >= 4, (r.wgs_per_chain, r.lane_group)
    assert np.array_equal(o["num_steps"][:, :2], r.num_steps[:, :2]), (o["num_steps"], r.num_steps)
    assert np.allclose(o["draws"][:, 0], r.draws[:, 0], atol=2e-3)
    ds.close()


# ---- the count models at their generators' default sizes: 100 sites x 52 visits ----
COUNT_GROUPS = [(1, None), (2, None), (8, None), (16, None), (4, 1)]


@pytest.mark.parametrize("g,gt", COUNT_GROUPS)
@pytest.mark.parametrize("name,mode", [("cop_default", None), ("cop_default", "constant"), ("cop_missing", "unoccupied")])
def test_occu_cop_with_groups(name, mode, g, gt, force_group):
    """occu_cop (occu_cop.py:197-255) on lane groups: K1 at its tolerance (2e-6 / 2e-5) and the oracle's first trees."""
    d = load_golden(name)
    kw = dict(model="occu_cop", fp_mode=mode, session_duration=d["session_duration"])
    od = oracle.OracleData(d["site_covs"], d["obs_covs"], d["obs"], **kw)
    ds = OccuDataset(d["site_covs"], d["obs_covs"], d["obs"], **kw)
    th = np.random.default_rng(3).uniform(-1.2, 1.2, size=(4, od.D)).astype(np.float32).astype(np.float64)
    if mode:
        th[0, -1], th[1, -1] = -5.0, 0.7
    Uo, Go = od.potential_grad(th)
    force_group(g, gt)
    Ug, Gg = ds.logp_grad(th)
    assert np.max(np.abs(Ug - Uo) / np.abs(Uo)) <= 2e-6, (Ug, Uo)
    assert np.max(np.abs(Gg - Go) / np.max(np.abs(Go), axis=1, keepdims=True)) <= 2e-5
    o = oracle.nuts_run(od, 0, 4, num_chains=2, seed=3)
    r = ds.nuts(num_warmup=0, num_samples=4, num_chains=2, seed=3)
    assert r.lane_group[0] * r.lane_group[1] == g, r.lane_group
    assert np.array_equal(o["num_steps"][:, :2], r.num_steps[:, :2]), (o["num_steps"], r.num_steps)
    ds.close()


@pytest.mark.parametrize("g,gt", COUNT_GROUPS)
@pytest.mark.parametrize("name,K", [("nmix_default", 100), ("nmix_ref_test_3periods", 19)])
def test_nmixture_with_groups(name, K, g, gt, force_group):
    """nmixture (nmixture.py:150-220) on lane groups: the visit lanes fold the slope's sum, the sums over n are formed by every lane alike."""
    d = load_golden(name)
    kw = dict(model="nmixture", max_abundance=K)
    od = oracle.OracleData(d["site_covs"], d["obs_covs"], d["obs"], **kw)
    ds = OccuDataset(d["site_covs"], d["obs_covs"], d["obs"], **kw)
    th = np.random.default_rng(3).uniform(-1.0, 1.0, size=(4, od.D)).astype(np.float32).astype(np.float64)
    Uo, Go = od.potential_grad(th)
    force_group(g, gt)
    Ug, Gg = ds.logp_grad(th)
    assert np.max(np.abs(Ug - Uo) / np.abs(Uo)) <= 2e-6, (Ug, Uo)
    assert np.max(np.abs(Gg - Go) / np.max(np.abs(Go), axis=1, keepdims=True)) <= 2e-5
    o = oracle.nuts_run(od, 0, 4, num_chains=2, seed=3)
    r = ds.nuts(num_warmup=0, num_samples=4, num_chains=2, seed=3)
    assert r.lane_group[0] * r.lane_group[1] == g, r.lane_group
    assert np.array_equal(o["num_steps"][:, :2], r.num_steps[:, :2]), (o["num_steps"], r.num_steps)
    ds.close()


def test_a_launch_reports_the_environment_knobs_it_saw(monkeypatch):
    """bl_nuts_env_overrides: "" when no BIOLITH_HIP_* knob is set (the engine's own geometry), the knobs and their values otherwise --
    what bench.py prints as config.env_overrides (INTEGRATION.md "Environment variables")."""
    d = load_golden("default")
    ds = OccuDataset(d["site_covs"], d["obs_covs"], d["obs"])
    for k in [k for k in os.environ if k.startswith("BIOLITH_HIP_") and k != "BIOLITH_HIP_LIB"]:
        monkeypatch.delenv(k)
    r0 = ds.nuts(num_warmup=10, num_samples=10, num_chains=1, seed=0)
    assert r0.env_overrides == ""
    monkeypatch.setenv("BIOLITH_HIP_OCCU_G", "4")
    monkeypatch.setenv("BIOLITH_HIP_SINGLE", "0")
    r1 = ds.nuts(num_warmup=10, num_samples=10, num_chains=1, seed=0)
    assert r1.env_overrides == "BIOLITH_HIP_OCCU_G=4,BIOLITH_HIP_SINGLE=0" and r1.lane_group[0] * r1.lane_group[1] == 4
    monkeypatch.delenv("BIOLITH_HIP_OCCU_G")
    monkeypatch.delenv("BIOLITH_HIP_SINGLE")
    r2 = ds.nuts(num_warmup=10, num_samples=10, num_chains=1, seed=0)
    assert r2.env_overrides == "" and np.array_equal(r2.draws, r0.draws)
    ds.close()
