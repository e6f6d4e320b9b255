"""N-mixture model with random effects (biolith/models/nmixture.py:139-141, 166-172, 199-214): site_re_abu joins the abundance
predictor, site_re_det and obs_re the detection predictor.  theta = [beta, alpha, (log sds), (effects)].  The kernels
(re_kernel.hpp, kind 3) through the C-ABI (bl_dataset_create_nmix_re) against the float64 oracle: potential + gradient over every
coordinate, the first trees on shared streams (one and several workgroups per chain), the posterior, and the reference's own
three fit tests (nmixture.py:516-620)."""
import contextlib
import io

import numpy as np
import pytest

import oracle
from biolith_amd.engine import OccuDataset
from biolith_amd.models import nmixture, simulate_nmixture
from biolith_amd.utils import fit, predict
from conftest import load_golden, load_oracle_draws, posterior_parity

pytestmark = pytest.mark.gpu

CASES = [("nmix_ref_test", 9, True, False), ("nmix_ref_test", 30, False, True), ("nmix_ref_test_3periods", 19, True, True),
         ("nmix_small_2x2", 40, True, True), ("nmix_site_re", 0, True, False), ("nmix_both_re", 0, True, True)]
REF_TEST = dict(simulate_missing=True, deployment_days_per_site=70, session_duration=7, min_abundance=1.0,
                min_observation_rate=1.0, max_observation_rate=6.0)


def _pair(name, K, site, obs, **kw):
    g = load_golden(name)
    K = K or int(np.nanmax(g["obs"])) + 5
    kw = dict(model="nmixture", max_abundance=K, site_random_effects=site, obs_random_effects=obs, prior_site_re_sd=0.8, prior_obs_re_sd=1.2, **kw)
    return (g, oracle.OracleData(g["site_covs"], g["obs_covs"], g["obs"], **kw), OccuDataset(g["site_covs"], g["obs_covs"], g["obs"], **kw))


@pytest.mark.parametrize("name,K,site,obs", CASES)
def test_nmix_re_logp_grad_parity(name, K, site, obs):
    """float32 kernel vs float64 oracle over every coordinate (the random-effects models' tolerances: 2e-6 / 2e-5)."""
    _, od, ds = _pair(name, K, site, obs)
    assert ds.D == od.D
    th = np.random.default_rng(4).uniform(-0.8, 0.8, size=(3, od.D)).astype(np.float32).astype(np.float64)
    Uo, Go = od.potential_grad(th)
    Ug, Gg = ds.logp_grad(th)
    assert np.all(np.isfinite(Ug)) and np.all(np.isfinite(Gg))
    assert np.max(np.abs(Ug - Uo) / np.abs(Uo)) <= 2e-6, (Ug, Uo)
    assert np.max(np.abs(Gg - Go)) <= 2e-5 * np.max(np.abs(Go)), np.max(np.abs(Gg - Go)) / np.max(np.abs(Go))


@pytest.mark.parametrize("n_sites", [1, 2, 65, 513])
def test_nmix_re_ragged_site_counts(n_sites):
    rng = np.random.default_rng(n_sites)
    X = rng.normal(size=(n_sites, 2)) * 0.5; W = rng.normal(size=(n_sites, 2, 3, 2)) * 0.5
    Nn = rng.poisson(2.0, size=(n_sites, 2, 1))
    Y = rng.binomial(Nn, 0.4, size=(n_sites, 2, 3)).astype(float)[None]
    Y[rng.uniform(size=Y.shape) < 0.1] = np.nan
    kw = dict(model="nmixture", max_abundance=25, site_random_effects=True, obs_random_effects=True)
    od, ds = oracle.OracleData(X, W, Y, **kw), OccuDataset(X, W, Y, **kw)
    th = rng.uniform(-0.8, 0.8, size=(2, od.D)).astype(np.float32).astype(np.float64)
    Uo, Go = od.potential_grad(th)
    for staged in (True, False):   # rows in LDS / read from device memory
        Ug, Gg = ds.logp_grad(th, staged=staged)
        assert np.max(np.abs(Ug - Uo) / np.abs(Uo)) <= 2e-6
        assert np.max(np.abs(Gg - Go)) <= 2e-5 * np.max(np.abs(Go))


@pytest.mark.parametrize("k", [1, 3, 16])
@pytest.mark.parametrize("name,K,site,obs", CASES[:3])
def test_nmix_re_first_transitions_match_oracle(name, K, site, obs, k):
    _, od, ds = _pair(name, K, site, obs)
    o = oracle.nuts_run(od, 0, 4, num_chains=2, seed=3)
    r = ds.nuts(num_warmup=0, num_samples=4, num_chains=2, seed=3, wgs_per_chain=k)
    assert np.array_equal(o["num_steps"][:, :2], r.num_steps[:, :2]), (o["num_steps"], r.num_steps)
    assert np.allclose(o["draws"][:, 0], r.draws[:, 0], atol=5e-3)


def test_nmix_re_adaptation_and_next_tree_match_oracle():
    _, od, ds = _pair("nmix_ref_test", 9, True, False)
    o = oracle.nuts_run(od, 8, 3, num_chains=2, seed=5)
    r = ds.nuts(num_warmup=8, num_samples=3, num_chains=2, seed=5)
    assert np.allclose(o["step_size"], r.step_size, rtol=2e-3)
    assert np.array_equal(o["num_steps"][:, :1], r.num_steps[:, :1])


def test_nmix_re_posterior_matches_oracle():
    _, od, ds = _pair("nmix_site_re", 0, True, False)
    G = od.Ks + od.Ko + 2
    # (the oracle's 4 x (500 + 1000), 76 s of CPU, is a committed fixture: tests/golden/make_oracle_posterior_draws.py)
    od_draws = load_oracle_draws("nmix_site_re", D=od.D, warmup=500, samples=1000)
    r = ds.nuts(num_warmup=500, num_samples=1000, num_chains=4, seed=50)
    # the fixed effects at SURVEY 8c's tolerances; log site_re_sd (the centred parameterisation's funnel: test_gpu_re.py) in mean
    posterior_parity(r.draws[:, :, :G], od_draws[:, :, :G])
    sg, so = r.draws[:, :, G:G + 1].astype(np.float64), od_draws[:, :, G:G + 1]
    mcse = np.sqrt(sg.var() / oracle.effective_sample_size(sg)[0] + so.var() / oracle.effective_sample_size(so)[0])
    assert abs(sg.mean() - so.mean()) <= 4 * mcse, (sg.mean(), so.mean(), mcse)


def _fit(data, **kw):
    return fit(nmixture, **data, max_abundance=int(np.nanmax(data["obs"])), num_chains=1, timeout=600, **kw)


def test_reference_nmixture_site_random_effects():
    """nmixture.py:516-551."""
    with contextlib.redirect_stdout(io.StringIO()):
        data, truth = simulate_nmixture(**{**REF_TEST, "deployment_days_per_site": 140}, site_random_effects=True, obs_random_effects=False)
    res = _fit(data, site_random_effects=True, obs_random_effects=False, num_samples=300)
    s = res.samples
    assert "site_re_sd" in s and "site_re_abu" in s and "site_re_det" in s
    assert s["site_re_sd"].mean() > 0
    assert s["site_re_abu"].shape == (300, 100, 1) and s["abundance"].shape == (300, 1, 100, 1)
    assert np.allclose(s["abundance"].mean(), truth["abundance"].mean(), rtol=0.25)
    # predict(): N_i and the counts drawn with the effects in both predictors (nmixture.py:181-220)
    K = int(np.nanmax(data["obs"]))
    pred = predict(nmixture, res.mcmc, **data, site_random_effects=True, max_abundance=K, num_samples=300)
    J = data["obs"].shape[3]
    assert pred["N_i"].shape == (300, 1, 100, 1) and pred["y"].shape == (300, J, 1, 100, 1)
    assert np.allclose(pred["abundance"], s["abundance"], rtol=1e-5)
    assert pred["N_i"].max() <= K and pred["y"].max() <= K
    # E[y | N, p] = N p: the predicted counts against the latent draws and the detection probabilities they were made from
    expect = (pred["N_i"][:, None].astype(np.float64) * pred["prob_detection"]).mean()
    assert abs(pred["y"].mean() - expect) < 0.05 * max(expect, 1.0)


def test_reference_nmixture_obs_random_effects():
    """nmixture.py:554-585."""
    with contextlib.redirect_stdout(io.StringIO()):
        data, truth = simulate_nmixture(**REF_TEST)
    res = _fit(data, obs_random_effects=True, num_samples=300)
    s = res.samples
    assert "obs_re_sd" in s and "obs_re" in s
    assert s["obs_re_sd"].mean() > 0
    assert s["obs_re"].shape == (300, 10, 1, 100, 1)
    assert np.allclose(s["abundance"].mean(), truth["abundance"].mean(), rtol=0.25)


def test_reference_nmixture_combined_random_effects():
    """nmixture.py:588-620."""
    with contextlib.redirect_stdout(io.StringIO()):
        data, _ = simulate_nmixture(**REF_TEST)
    res = _fit(data, site_random_effects=True, obs_random_effects=True, num_warmup=10, num_samples=10)
    for k in ("site_re_sd", "site_re_abu", "site_re_det", "obs_re_sd", "obs_re"):
        assert k in res.samples


def test_nmix_re_rejects_several_species():
    with contextlib.redirect_stdout(io.StringIO()):
        data, _ = simulate_nmixture(n_species=2, n_sites=30, random_seed=1)
    with pytest.raises(NotImplementedError):
        fit(nmixture, **data, site_random_effects=True, num_chains=1, num_samples=5, num_warmup=5)
