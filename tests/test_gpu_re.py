"""Random-effects occupancy model (biolith/models/occu.py:170-173, 191-196, 215-218) through the C-ABI against the CPU
oracle: potential + gradient over all D coordinates, identical first trees from the same RNG streams, posterior."""
import numpy as np
import pytest

import oracle
from biolith_amd.engine import OccuDataset
from biolith_amd.evaluation import split_gelman_rubin
from conftest import load_golden

pytestmark = pytest.mark.gpu

CASES = [("small_3x3", True, False, (1.0, 1.0)), ("small_3x3", False, True, (1.0, 0.5)), ("missing", True, True, (0.7, 2.0)),
         ("default", True, True, (1.0, 1.0))]


def _pair(name, site, obs, scales):
    g = load_golden(name)
    kw = dict(site_random_effects=site, obs_random_effects=obs, prior_site_re_sd=scales[0], prior_obs_re_sd=scales[1])
    return (g, oracle.OracleData(g["site_covs"], g["obs_covs"], g["obs"], model="occu_re", **kw),
            OccuDataset(g["site_covs"], g["obs_covs"], g["obs"], model="occu_re", **kw))


@pytest.mark.parametrize("name,site,obs,scales", CASES)
def test_re_logp_grad_parity(name, site, obs, scales):
    """float32 kernel vs float64 oracle over every coordinate: |dU|/|U| <= 2e-6, max|dgrad| <= 2e-5 max|grad|."""
    _, od, ds = _pair(name, site, obs, scales)
    assert ds.D == od.D
    th = np.random.default_rng(4).uniform(-1.2, 1.2, size=(3, od.D)).astype(np.float32).astype(np.float64)
    Uo, Go = od.potential_grad(th)
    Ug, Gg = ds.logp_grad(th)
    assert np.all(np.isfinite(Ug)) and np.all(np.isfinite(Gg))
    assert np.max(np.abs(Ug - Uo) / np.abs(Uo)) <= 2e-6, (Ug, Uo)
    assert np.max(np.abs(Gg - Go)) <= 2e-5 * np.max(np.abs(Go)), np.max(np.abs(Gg - Go))


@pytest.mark.parametrize("name,site,obs,scales", CASES[:3])
def test_re_first_transitions_match_oracle(name, site, obs, scales):
    _, od, ds = _pair(name, site, obs, scales)
    o = oracle.nuts_run(od, 0, 4, num_chains=2, seed=3)
    r = ds.nuts(num_warmup=0, num_samples=4, num_chains=2, seed=3)
    assert np.array_equal(o["num_steps"][:, :2], r.num_steps[:, :2]), (o["num_steps"], r.num_steps)
    assert np.allclose(o["draws"][:, 0], r.draws[:, 0], atol=5e-3)


@pytest.mark.parametrize("k", [2, 5, 32])
@pytest.mark.parametrize("name,site,obs,scales", CASES[:3])
def test_re_sliced_chains_build_the_same_trees(name, site, obs, scales, k):
    """A chain spread over k workgroups (each a slice of the sites, partial sums exchanged through device memory) takes
    the same decisions as the one-workgroup form and as the oracle: same trees, same draws to float32 accuracy."""
    _, od, ds = _pair(name, site, obs, scales)
    o = oracle.nuts_run(od, 0, 4, num_chains=2, seed=3)
    r1 = ds.nuts(num_warmup=0, num_samples=4, num_chains=2, seed=3, wgs_per_chain=1)
    rk = ds.nuts(num_warmup=0, num_samples=4, num_chains=2, seed=3, wgs_per_chain=k)
    assert 1 < rk.wgs_per_chain <= min(k, od.N) and r1.wgs_per_chain == 1   # (no empty slice: 300 sites / 32 -> 30 workgroups)
    assert np.array_equal(o["num_steps"][:, :2], rk.num_steps[:, :2]), (o["num_steps"], rk.num_steps)
    assert np.array_equal(r1.num_steps[:, :2], rk.num_steps[:, :2])
    assert np.allclose(o["draws"][:, 0], rk.draws[:, 0], atol=5e-3)
    assert np.allclose(r1.draws[:, :2], rk.draws[:, :2], atol=2e-3)
    # a warmed-up run on the sliced form: finite, adapted, every coordinate written
    w = ds.nuts(num_warmup=60, num_samples=20, num_chains=2, seed=5, wgs_per_chain=k)
    assert np.all(np.isfinite(w.draws)) and np.all(w.step_size > 1e-3) and np.all(w.inv_mass > 0)
    assert np.all(np.abs(w.draws).max(axis=(0, 1)) > 0)


def test_re_warmup_trajectory_matches_oracle():
    """Same streams, same adaptation: step sizes and tree sizes of the first warmup transitions agree -- closely between
    the one-workgroup and the sliced form (float32 partial sums differ in the last bits only), loosely with the float64
    oracle (trajectories separate chaotically after a few dozen transitions)."""
    _, od, ds = _pair("small_3x3", True, False, (1.0, 1.0))
    # (six chains, six transitions each: the last bits of a float32 sum can tip a multinomial pick that sits on its threshold, and the two
    # forms part there -- after ten transitions two chains in six had; most stay together through warm-up and draws)
    same = []
    for seed in (11, 12, 13):
        a = ds.nuts(num_warmup=4, num_samples=2, num_chains=2, seed=seed, wgs_per_chain=1)
        b = ds.nuts(num_warmup=4, num_samples=2, num_chains=2, seed=seed, wgs_per_chain=3)
        for c in range(2):
            same.append(bool(np.allclose(a.step_size[c], b.step_size[c], rtol=2e-3) and np.array_equal(a.n_leapfrog[c], b.n_leapfrog[c])
                             and np.allclose(a.draws[c], b.draws[c], atol=5e-3)))
    assert sum(same) >= 4, same
    o = oracle.nuts_run(od, 30, 5, num_chains=1, seed=11)
    r = ds.nuts(num_warmup=30, num_samples=5, num_chains=1, seed=11)
    assert abs(np.log(r.step_size[0] / o["step_size"][0])) < 0.6
    assert abs(int(r.n_leapfrog.sum()) - int(o["n_leapfrog"].sum())) <= 0.4 * int(o["n_leapfrog"].sum())
    # mass-matrix adaptation on the sliced form: a full schedule, compared with the one-workgroup form in distribution
    a = ds.nuts(num_warmup=150, num_samples=50, num_chains=2, seed=2, wgs_per_chain=1)
    b = ds.nuts(num_warmup=150, num_samples=50, num_chains=2, seed=2, wgs_per_chain=3)
    assert np.all(np.abs(np.log(b.step_size / a.step_size)) < 0.5)
    assert 0.5 < np.median(b.inv_mass / a.inv_mass) < 2.0


def test_re_posterior_matches_oracle():
    _, od, ds = _pair("small_3x3", True, False, (1.0, 1.0))
    o = oracle.nuts_run(od, 400, 400, num_chains=4, seed=0)
    r = ds.nuts(num_warmup=400, num_samples=400, num_chains=4, seed=50)
    G = od.Ks + od.Ko + 3
    fg, fo = r.draws[:, :, :G].reshape(-1, G).astype(np.float64), o["draws"][:, :, :G].reshape(-1, G)
    ess_g = np.array([oracle.effective_sample_size(r.draws[:, :, k:k + 1].astype(np.float64))[0] for k in range(G)])
    ess_o = np.array([oracle.effective_sample_size(o["draws"][:, :, k:k + 1])[0] for k in range(G)])
    mcse = np.sqrt(fg.var(0) / ess_g + fo.var(0) / ess_o)
    assert np.all(np.abs(fg.mean(0) - fo.mean(0)) <= 4 * mcse), (fg.mean(0) - fo.mean(0), mcse)
    ratio = (fg.std(0) / fo.std(0))[:G - 1]
    assert np.all((ratio > 0.75) & (ratio < 1.33)), ratio
    # the fixed effects mix; log site_re_sd sits at the neck of the centred parameterisation's funnel (the reference's own
    # parameterisation) and is only required to agree with the oracle's draws in mean and spread (above)
    assert split_gelman_rubin(r.draws[:, :, :G - 1]).max() < 1.1
    assert r.diverging.mean() < 0.05


# ---- the reference's own tests of these options (biolith/models/occu.py:770-863), through fit() ----
def test_site_random_effects_like_reference():
    from biolith_amd.evaluation import lppd
    from biolith_amd.models import occu, simulate
    from biolith_amd.utils import fit, predict

    data, true_params = simulate(site_random_effects=True, obs_random_effects=False, deployment_days_per_site=7000)
    results = fit(occu, **data, num_chains=1, num_samples=500, timeout=600)
    l = lppd(occu, predict(occu, results.mcmc, **data), **data)
    results_re = fit(occu, **data, site_random_effects=True, obs_random_effects=False, num_chains=1, num_samples=500, timeout=600)
    posterior_samples_re = predict(occu, results_re.mcmc, **data, site_random_effects=True, obs_random_effects=False)
    l_re = lppd(occu, posterior_samples_re, **data)
    assert l_re >= 0.95 * l
    assert results_re.samples["site_re_sd"].shape == (500,)
    assert results_re.samples["site_re_occ"].shape == (500, 100, 1) and results_re.samples["site_re_det"].shape == (500, 100, 1)
    assert results_re.samples["site_re_sd"].mean() > 0
    assert np.allclose(results_re.samples["psi"].mean(), true_params["z"].mean(), atol=0.15)
    # the simulated effects are recovered: posterior means correlate with the truth where the data are this rich
    # (at the occupied sites -- an unoccupied site says nothing about its detection effect)
    est = results_re.samples["site_re_det"].mean(0)[:, 0]
    occupied = true_params["z"].reshape(-1) == 1
    assert np.corrcoef(est[occupied], true_params["site_re_det"].reshape(-1)[occupied])[0, 1] > 0.8


def test_obs_random_effects_like_reference():
    from biolith_amd.models import occu, simulate
    from biolith_amd.utils import fit

    data, true_params = simulate(simulate_missing=True)
    results = fit(occu, **data, obs_random_effects=True, num_chains=1, num_samples=500, timeout=600)
    assert results.samples["obs_re_sd"].shape == (500,) and results.samples["obs_re"].shape == (500, 52, 1, 100, 1)
    assert results.samples["obs_re_sd"].mean() > 0
    assert np.allclose(results.samples["psi"].mean(), true_params["z"].mean(), atol=0.15)


def test_combined_random_effects_like_reference():
    from biolith_amd.models import occu, simulate
    from biolith_amd.utils import fit

    data, true_params = simulate(simulate_missing=True)
    results = fit(occu, **data, site_random_effects=True, obs_random_effects=True, num_chains=1, num_samples=500, timeout=600)
    for k in ("site_re_sd", "site_re_occ", "site_re_det", "obs_re_sd", "obs_re"):
        assert k in results.samples
    assert np.allclose(results.samples["psi"].mean(), true_params["z"].mean(), atol=0.15)
    r = results.mcmc.result
    print("combined RE: D", r.draws.shape[-1], "kernel ms", r.kernel_ms, "leapfrogs", int(r.n_leapfrog.sum()),
          "us/leapfrog", 1e3 * r.kernel_ms / max(int(r.n_leapfrog.sum()), 1))


def test_re_chains_sharded_over_devices_reproduce_the_single_launch():
    """fit(devices=[0, 0]): two launches of two chains each (chain_offset 0 and 2) draw the same streams as one launch of
    four -- the random-effects streams are D + 2 per chain, offset by the chain's global index."""
    from biolith_amd.models import occu, simulate
    from biolith_amd.utils import fit

    data, _ = simulate(simulate_missing=True, n_sites=60, deployment_days_per_site=70)
    kw = dict(site_random_effects=True, obs_random_effects=True, num_chains=4, num_samples=20, num_warmup=20, random_seed=4)
    one = fit(occu, **data, **kw)
    two = fit(occu, **data, **kw, devices=[0, 0])
    for k in ("cov_state_0", "site_re_sd", "obs_re_sd"):
        assert np.allclose(one.samples[k], two.samples[k], atol=1e-5), k
    assert np.allclose(one.samples["obs_re"], two.samples["obs_re"], atol=1e-4)


@pytest.mark.parametrize("ks,ko,periods", [(0, 0, 1), (4, 4, 2), (2, 1, 3)])
def test_re_covariate_counts_and_periods(ks, ko, periods):
    """Dimension extremes (no covariates at all, four per side) and several periods sharing one site effect: potential and
    gradient against the oracle over every coordinate, identical first trees, on the one-workgroup and on a sliced chain."""
    rng = np.random.default_rng(10 * ks + ko + periods)
    N, J = 70, 6
    X = rng.normal(size=(N, ks)).astype(np.float32)
    W = rng.normal(size=(N, periods, J, ko)).astype(np.float32)
    Y = (rng.uniform(size=(1, N, periods, J)) < 0.35).astype(np.float32)
    Y[0, rng.integers(0, N, 25), rng.integers(0, periods, 25), rng.integers(0, J, 25)] = np.nan
    if ko:
        W[rng.integers(0, N, 5), 0, rng.integers(0, J, 5), 0] = np.nan
    kw = dict(model="occu_re", site_random_effects=True, obs_random_effects=True, prior_site_re_sd=0.8, prior_obs_re_sd=1.3)
    od, ds = oracle.OracleData(X, W, Y, **kw), OccuDataset(X, W, Y, **kw)
    assert ds.D == od.D == ks + ko + 4 + 2 * N + N * periods * J
    th = rng.uniform(-1.0, 1.0, size=(2, od.D)).astype(np.float32).astype(np.float64)
    Uo, Go = od.potential_grad(th)
    Ug, Gg = ds.logp_grad(th)
    assert np.max(np.abs(Ug - Uo) / np.abs(Uo)) <= 2e-6 and np.max(np.abs(Gg - Go)) <= 2e-5 * np.max(np.abs(Go))
    o = oracle.nuts_run(od, 0, 3, num_chains=2, seed=8)
    for k in (1, 4):
        r = ds.nuts(num_warmup=0, num_samples=3, num_chains=2, seed=8, wgs_per_chain=k)
        assert np.array_equal(o["num_steps"][:, :2], r.num_steps[:, :2]), (k, o["num_steps"], r.num_steps)
        assert np.allclose(o["draws"][:, 0], r.draws[:, 0], atol=5e-3)


def test_re_timeout_stops_every_workgroup_of_the_chain():
    """fit(timeout=...) on the sliced sampler: the abort request travels inside the exchanged sums, so all workgroups of a
    chain leave at the same leapfrog (none is left spinning on the others) and the handle is usable afterwards."""
    import time

    from biolith_amd.models import occu, simulate
    from biolith_amd.utils import fit
    from biolith_amd.utils.misc import TimeoutException

    data, _ = simulate(simulate_missing=True)
    t0 = time.time()
    with pytest.raises(TimeoutException):
        fit(occu, **data, obs_random_effects=True, num_chains=2, num_samples=10, num_warmup=10 ** 8, timeout=1)
    assert time.time() - t0 < 30
    res = fit(occu, **data, obs_random_effects=True, num_chains=2, num_samples=20, num_warmup=20)
    assert np.all(np.isfinite(res.samples["obs_re_sd"]))


@pytest.mark.parametrize("n_sites,site,obs", [(10000, True, False), (2000, True, True)])
def test_re_large_sizes_against_the_oracle(n_sites, site, obs):
    """The bench-sized datasets (20 009 / 24 010 coordinates; 32 workgroups per chain, rows too large for one workgroup's
    LDS at 10 000 sites): potential and gradient over every coordinate, and the first trees, against the oracle."""
    import contextlib
    import io

    from biolith_amd.models import simulate

    with contextlib.redirect_stdout(io.StringIO()):
        d, _ = simulate(n_sites=n_sites, n_site_covs=3, n_obs_covs=3, deployment_days_per_site=70, session_duration=7,
                        site_random_effects=site, obs_random_effects=obs, random_seed=0)
    kw = dict(model="occu_re", site_random_effects=site, obs_random_effects=obs)
    od, ds = oracle.OracleData(d["site_covs"], d["obs_covs"], d["obs"], **kw), OccuDataset(d["site_covs"], d["obs_covs"], d["obs"], **kw)
    assert ds.D == od.D == 8 + site + obs + (2 * n_sites if site else 0) + (n_sites * 10 if obs else 0)
    th = np.random.default_rng(0).uniform(-1.0, 1.0, size=(2, od.D)).astype(np.float32).astype(np.float64)
    Uo, Go = od.potential_grad(th)
    Ug, Gg = ds.logp_grad(th)
    assert np.max(np.abs(Ug - Uo) / np.abs(Uo)) <= 2e-6, (Ug, Uo)
    assert np.max(np.abs(Gg - Go)) <= 2e-5 * np.max(np.abs(Go))
    o = oracle.nuts_run(od, 0, 2, num_chains=2, seed=1)
    r = ds.nuts(num_warmup=0, num_samples=2, num_chains=2, seed=1)
    assert r.wgs_per_chain >= 16
    assert np.array_equal(o["num_steps"], r.num_steps), (o["num_steps"], r.num_steps)
    assert np.allclose(o["draws"], r.draws, atol=5e-3)


def _with_env(env, fn):
    import os
    old = {k: os.environ.get(k) for k in env}
    os.environ.update({k: str(v) for k, v in env.items()})
    try:
        return fn()
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


@pytest.mark.parametrize("what", ["re_cap4", "re_cap16", "cs"])
def test_re_kernel_instantiations_agree_bit_for_bit(what):
    """bl_re_nuts_kernel<MK, KIND, LROWS, LT> is compiled for the covariate capacity, the model kind, rows in LDS or in device memory,
    and none / five / thirty sampler vectors in LDS.  Where the data live changes no arithmetic: all six (rows, tier) forms of a
    capacity and kind give bit-identical draws (the launch reports which form ran)."""
    rng = np.random.default_rng(5)
    if what == "cs":
        g = load_golden("cs_small_2x2")
        kw = dict(model="occu_cs", prior_mu=((0.5, 8.0), (1.0, 12.0)), prior_sigma=((5.0, 1.0), (3.0, 0.5)))
        ds = OccuDataset(g["site_covs"], g["obs_covs"], g["obs"], **kw)
        init = rng.uniform(-1, 1, size=(2, ds.D))
        init[:, -4:] = np.array([0.3, 1.8, 1.9, 1.2])
    else:
        ks, ko = (3, 2) if what == "re_cap4" else (6, 9)
        N, J = 90, 6
        X = rng.normal(size=(N, ks)).astype(np.float32)
        W = rng.normal(size=(N, 1, J, ko)).astype(np.float32)
        Y = (rng.uniform(size=(1, N, 1, J)) < 0.4).astype(np.float32)
        ds = OccuDataset(X, W, Y, model="occu_re", site_random_effects=True, obs_random_effects=True)
        init = rng.uniform(-0.5, 0.5, size=(2, ds.D))
    run = lambda: ds.nuts(num_warmup=30, num_samples=10, num_chains=2, seed=2, init_theta=init, wgs_per_chain=3)
    ref = _with_env({}, run)
    assert ref.lds_staged and ref.lds_vector_tier == 2 and np.all(np.isfinite(ref.draws))
    for rows in (1, 0):
        for tier in (2, 1, 0):
            r = _with_env(dict(BIOLITH_HIP_RE_LDS_ROWS=rows, BIOLITH_HIP_RE_LDS_TIER=tier), run)
            assert (r.lds_staged, r.lds_vector_tier) == (bool(rows), tier)
            assert np.array_equal(r.draws, ref.draws) and np.array_equal(r.num_steps, ref.num_steps), (rows, tier)
            assert np.array_equal(r.step_size, ref.step_size) and np.array_equal(r.inv_mass, ref.inv_mass)


@pytest.mark.parametrize("site,obs", [(True, False), (False, True), (True, True)])
def test_re_effects_as_compile_time_facts_change_no_bit(site, obs):
    """bl_re_nuts_kernel<4, 0, true, 2, EFF>: the bench form with the model's effects (and, for site effects alone, its one period) as
    compile-time facts runs the same arithmetic as the general kernel (BIOLITH_HIP_RE_EFF=0): bit-identical draws, trees, adaptation."""
    rng = np.random.default_rng(17)
    N, J = 150, 7
    X = rng.normal(size=(N, 3)).astype(np.float32)
    W = rng.normal(size=(N, 1, J, 2)).astype(np.float32)
    Y = (rng.uniform(size=(1, N, 1, J)) < 0.4).astype(np.float32)
    ds = OccuDataset(X, W, Y, model="occu_re", site_random_effects=site, obs_random_effects=obs)
    init = rng.uniform(-0.5, 0.5, size=(2, ds.D))
    run = lambda: ds.nuts(num_warmup=40, num_samples=20, num_chains=2, seed=4, init_theta=init, wgs_per_chain=2)
    fast = _with_env({}, run)
    assert fast.lds_staged and fast.lds_vector_tier == 2 and np.all(np.isfinite(fast.draws))
    eff = (1 if site else 0) | (2 if obs else 0)
    assert fast.kernel_name == f"bl_re_nuts_kernel<4, 0, true, 2, {eff + 4 if eff == 1 else eff}>"  # (what ran, as rocprofv3 names it)
    for knob in ("0", "1"):  # the general kernel; the facts without the one-period one
        r = _with_env(dict(BIOLITH_HIP_RE_EFF=knob), run)
        assert r.kernel_name == f"bl_re_nuts_kernel<4, 0, true, 2, {0 if knob == '0' else eff}>"
        assert np.array_equal(r.draws, fast.draws) and np.array_equal(r.num_steps, fast.num_steps), knob
        assert np.array_equal(r.step_size, fast.step_size) and np.array_equal(r.inv_mass, fast.inv_mass)
    ds.close()


def test_re_site_pass_wave_classes_build_the_same_trees():
    """300 sites in one workgroup fill 4.7 waves: four waves take a site per lane, the other four share the remaining 44 sites two
    lanes per site (BlReSiteMap).  Same trees and draws (to float32 summation order) as with one class, and as the oracle."""
    rng = np.random.default_rng(11)
    N, J = 300, 8
    X = rng.normal(size=(N, 2)).astype(np.float32)
    W = rng.normal(size=(N, 1, J, 2)).astype(np.float32)
    Y = (rng.uniform(size=(1, N, 1, J)) < 0.4).astype(np.float32)
    kw = dict(model="occu_re", site_random_effects=True, obs_random_effects=True)
    od, ds = oracle.OracleData(X, W, Y, **kw), OccuDataset(X, W, Y, **kw)
    run = lambda: ds.nuts(num_warmup=0, num_samples=4, num_chains=2, seed=3, wgs_per_chain=1)
    two, one = _with_env({}, run), _with_env(dict(BIOLITH_HIP_RE_NO_SPLIT=1), run)
    o = oracle.nuts_run(od, 0, 4, num_chains=2, seed=3)
    assert np.array_equal(o["num_steps"][:, :2], two.num_steps[:, :2]) and np.array_equal(one.num_steps[:, :2], two.num_steps[:, :2])
    assert np.allclose(o["draws"][:, 0], two.draws[:, 0], atol=5e-3) and np.allclose(one.draws[:, :2], two.draws[:, :2], atol=2e-3)


SPECIES_CASES = [(40, 2, 5, 2, 1, 3, True, True), (300, 1, 6, 3, 3, 2, True, False), (70, 1, 6, 6, 9, 2, False, True)]


def _species_data(N, T, J, Ks, Ko, S, seed=0):
    rng = np.random.default_rng(seed)
    X = rng.normal(size=(N, Ks)).astype(np.float32)
    W = rng.normal(size=(N, T, J, Ko)).astype(np.float32)
    Y = (rng.uniform(size=(S, N, T, J)) < 0.4).astype(np.float32)
    Y[1, 3, 0, 1] = np.nan
    W[5, 0, 2, 0] = np.nan
    return rng, X, W, Y


@pytest.mark.parametrize("N,T,J,Ks,Ko,S,site,obs", SPECIES_CASES)
def test_re_several_species_share_the_sds(N, T, J, Ks, Ko, S, site, obs):
    """Random effects inside the species plate, their sds outside it (occu.py:170-173, 182-196): ONE chain over
    [species' beta, alpha | log sds | site_re_occ [S][N] | site_re_det [S][N] | obs_re [S][N][T][J]].  A chain's workgroups are S
    groups, each slicing the sites of its species.  Potential and gradient over every coordinate against the oracle, and -- from a
    start in the bulk -- the same step sizes after 8 adaptation steps and the same next tree and draw, on S, 2 S and up to 32 workgroups."""
    rng, X, W, Y = _species_data(N, T, J, Ks, Ko, S)
    kw = dict(model="occu_re", site_random_effects=site, obs_random_effects=obs, prior_site_re_sd=0.7, prior_obs_re_sd=1.3)
    od, ds = oracle.OracleData(X, W, Y, **kw), OccuDataset(X, W, Y, **kw)
    V = T * J
    assert ds.D == od.D == S * (Ks + Ko + 2) + int(site) + int(obs) + S * ((2 * N if site else 0) + (N * V if obs else 0))
    th = rng.uniform(-1, 1, size=(2, od.D)).astype(np.float32).astype(np.float64)
    Uo, Go = od.potential_grad(th)
    Ug, Gg = ds.logp_grad(th)
    assert np.max(np.abs(Ug - Uo) / np.abs(Uo)) <= 2e-6 and np.max(np.abs(Gg - Go)) <= 2e-5 * np.max(np.abs(Go))
    init = rng.uniform(-0.15, 0.15, size=(2, od.D)).astype(np.float32).astype(np.float64)
    o = oracle.nuts_run(od, 8, 4, num_chains=2, seed=3, init=init)
    assert o["num_steps"].max() > 7
    for k in (S, 2 * S, 32):
        r = ds.nuts(num_warmup=8, num_samples=4, num_chains=2, seed=3, wgs_per_chain=k, init_theta=init)
        assert r.wgs_per_chain % S == 0 and r.wgs_per_chain <= 32
        # (nine transitions in: the adapted step sizes and the first kept tree and draw; a float32 / float64 pair parts for good at the
        # first very long tree -- here the 415-step one of transition 11)
        assert np.array_equal(o["num_steps"][:, :1], r.num_steps[:, :1]), (k, o["num_steps"], r.num_steps)
        assert np.allclose(o["step_size"], r.step_size, rtol=2e-3) and np.allclose(o["draws"][:, 0], r.draws[:, 0], atol=5e-3)


def test_re_several_species_through_fit_and_predict():
    """fit(occu, site_random_effects=True, obs_random_effects=True) with three species: the reference's site shapes (species plate
    last, occu.py:182), shared sds, finite deterministic sites; predict() draws from them; joint_species=False does not apply."""
    from biolith_amd.models import occu, simulate
    from biolith_amd.utils import fit, predict

    data, _ = simulate(n_species=3, n_sites=60, deployment_days_per_site=56, site_random_effects=True, obs_random_effects=True, random_seed=3)
    kw = dict(site_random_effects=True, obs_random_effects=True, num_chains=2, num_samples=30, num_warmup=30, random_seed=1)
    res = fit(occu, **data, **kw)
    s = res.samples
    n, N, T, J = 60, 60, data["obs"].shape[2], data["obs"].shape[3]
    assert s["site_re_sd"].shape == (n,) and s["obs_re_sd"].shape == (n,)
    assert s["site_re_occ"].shape == s["site_re_det"].shape == (n, N, 3) and s["obs_re"].shape == (n, J, T, N, 3)
    assert s["cov_state_0"].shape == (n, 3) and s["psi"].shape == (n, T, N, 3) and s["prob_detection"].shape == (n, J, T, N, 3)
    assert all(np.all(np.isfinite(s[k])) for k in ("site_re_sd", "obs_re_sd", "site_re_occ", "obs_re", "psi", "prob_detection"))
    assert np.all(s["site_re_sd"] > 0) and res.mcmc.result.draws.shape[2] == 3 * 4 + 2 + 3 * (2 * N + N * T * J)
    # the species' effects are their own coordinates (not copies of one another)
    assert not np.allclose(s["site_re_occ"][..., 0], s["site_re_occ"][..., 1])
    pp = predict(occu, res.mcmc, **data, site_random_effects=True, obs_random_effects=True)
    assert pp["psi"].shape == (n, T, N, 3) and pp["y"].shape[-1] == 3
    # psi of species 1 from predict() = psi of the fit (same draws, same effects)
    assert np.allclose(pp["psi"], s["psi"], atol=1e-5)
    with pytest.raises(NotImplementedError):
        fit(occu, **data, **kw, joint_species=False)
    # chains dealt over two shards (here the same GPU twice): the same draws as the single launch
    two = fit(occu, **data, **kw, devices=[0, 0])
    for k in ("cov_state_0", "site_re_sd", "obs_re_sd", "site_re_occ"):
        assert np.allclose(res.samples[k], two.samples[k], atol=1e-5), k


def test_re_several_species_posterior_matches_oracle():
    """Posterior of the fixed effects of two species and of the shared log site_re_sd, 4 x (400 + 400) on each side with independent
    streams: means within 4 Monte-Carlo standard errors of the oracle's, spreads within a third."""
    from biolith_amd.models import simulate
    import contextlib
    import io

    with contextlib.redirect_stdout(io.StringIO()):
        d, _ = simulate(n_species=2, n_sites=120, n_site_covs=1, n_obs_covs=1, deployment_days_per_site=56, session_duration=7,
                        site_random_effects=True, random_seed=5)
    kw = dict(model="occu_re", site_random_effects=True, obs_random_effects=False)
    od = oracle.OracleData(d["site_covs"], d["obs_covs"], d["obs"], **kw)
    ds = OccuDataset(d["site_covs"], d["obs_covs"], d["obs"], **kw)
    o = oracle.nuts_run(od, 400, 400, num_chains=4, seed=0)
    r = ds.nuts(num_warmup=400, num_samples=400, num_chains=4, seed=50)
    G = 2 * (od.Ks + od.Ko + 2) + 1
    fg, fo = r.draws[:, :, :G].reshape(-1, G).astype(np.float64), o["draws"][:, :, :G].reshape(-1, G)
    ess_g = np.array([oracle.effective_sample_size(r.draws[:, :, k:k + 1].astype(np.float64))[0] for k in range(G)])
    ess_o = np.array([oracle.effective_sample_size(o["draws"][:, :, k:k + 1])[0] for k in range(G)])
    mcse = np.sqrt(fg.var(0) / ess_g + fo.var(0) / ess_o)
    assert np.all(np.abs(fg.mean(0) - fo.mean(0)) <= 4 * mcse), (fg.mean(0) - fo.mean(0), mcse)
    ratio = (fg.std(0) / fo.std(0))[:G - 1]
    assert np.all((ratio > 0.75) & (ratio < 1.33)), ratio
    assert split_gelman_rubin(r.draws[:, :, :G - 1]).max() < 1.1 and r.diverging.mean() < 0.05
