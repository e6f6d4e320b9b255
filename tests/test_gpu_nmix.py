"""N-mixture kernel (MODEL 4) through the C-ABI (bl_dataset_create_nmix) against the float64 oracle, plus the
reference's own fit assertions (biolith/models/nmixture.py:372-449)."""
import numpy as np
import pytest

import oracle
from biolith_amd.engine import OccuDataset
from biolith_amd.models import nmixture, simulate_nmixture
from biolith_amd.utils import fit, predict
from conftest import PARITY_S, PARITY_W, load_golden, load_oracle_draws, posterior_parity

pytestmark = pytest.mark.gpu
U_RTOL, G_RTOL = 2e-6, 2e-5   # float32 per-term math, sums over N in float32

REF_TEST = dict(simulate_missing=True, deployment_days_per_site=70, session_duration=7, min_abundance=1.0,
                min_observation_rate=1.0, max_observation_rate=6.0)


def _pair(name, K, priors=((0.0, 1.0), (0.0, 1.0))):
    g = load_golden(name)
    kw = dict(model="nmixture", max_abundance=K)
    return (g, oracle.OracleData(g["site_covs"], g["obs_covs"], g["obs"], *priors, **kw),
            OccuDataset(g["site_covs"], g["obs_covs"], g["obs"], *priors, **kw))


@pytest.mark.parametrize("name,K", [("nmix_ref_test", 9), ("nmix_ref_test", 60), ("nmix_ref_test_3periods", 19),
                                     ("nmix_small_2x2", 40), ("nmix_default", 100), ("nmix_default", 127)])
def test_nmix_logp_grad_parity(name, K):
    _, od, ds = _pair(name, K, priors=((0.1, 1.5), (-0.2, 0.8)))
    th = np.random.default_rng(3).uniform(-1.0, 1.0, size=(5, od.D)).astype(np.float32).astype(np.float64)
    Uo, Go = od.potential_grad(th)
    Ug, Gg = ds.logp_grad(th)
    assert np.all(np.isfinite(Ug)) and np.all(np.isfinite(Gg))
    assert np.max(np.abs(Ug - Uo) / np.abs(Uo)) <= U_RTOL, (Ug, Uo)
    assert np.max(np.abs(Gg - Go) / np.max(np.abs(Go), axis=1, keepdims=True)) <= G_RTOL, np.abs(Gg - Go).max(1)


@pytest.mark.parametrize("n_sites", [1, 2, 65, 385, 1031])
def test_nmix_ragged_site_counts(n_sites):
    rng = np.random.default_rng(n_sites)
    X = rng.normal(size=(n_sites, 2)) * 0.5; W = rng.normal(size=(n_sites, 2, 3, 2)) * 0.5
    Nn = rng.poisson(2.0, size=(n_sites, 2, 1))
    Y = rng.binomial(Nn, 0.4, size=(n_sites, 2, 3)).astype(float)[None]
    Y[rng.uniform(size=Y.shape) < 0.1] = np.nan
    kw = dict(model="nmixture", max_abundance=25)
    od, ds = oracle.OracleData(X, W, Y, **kw), OccuDataset(X, W, Y, **kw)
    th = rng.uniform(-0.8, 0.8, size=(2, 6)).astype(np.float32).astype(np.float64)
    Uo, Go = od.potential_grad(th)
    Ug, Gg = ds.logp_grad(th)
    assert np.max(np.abs(Ug - Uo) / np.abs(Uo)) <= U_RTOL
    assert np.max(np.abs(Gg - Go)) <= G_RTOL * np.max(np.abs(Go))


def test_nmix_limits():
    g = load_golden("nmix_small_2x2")
    with pytest.raises(ValueError, match="below the largest count"):
        OccuDataset(g["site_covs"], g["obs_covs"], g["obs"], model="nmixture", max_abundance=5)
    with pytest.raises((RuntimeError, NotImplementedError, ValueError)):
        OccuDataset(g["site_covs"], g["obs_covs"], g["obs"], model="nmixture", max_abundance=128)


def test_nmix_first_transitions_match_oracle():
    _, od, ds = _pair("nmix_small_2x2", 40)
    o = oracle.nuts_run(od, 0, 5, num_chains=2, seed=3)
    r = ds.nuts(num_warmup=0, num_samples=5, num_chains=2, seed=3)
    assert np.array_equal(o["num_steps"][:, :3], r.num_steps[:, :3]), (o["num_steps"], r.num_steps)
    assert np.allclose(o["draws"][:, 0], r.draws[:, 0], atol=2e-3)


def test_nmix_posterior_matches_oracle():
    """The oracle's leg is a committed fixture (tests/golden/make_oracle_posterior_draws.py: the same call, 59 s of CPU -- two thirds of this
    test's 42 s on the GPU box until round 6); tests/test_nmix_cpu.py keeps the oracle itself honest."""
    _, od, ds = _pair("nmix_small_2x2", 40)
    o = load_oracle_draws("nmix_small_2x2", D=od.D, warmup=PARITY_W, samples=PARITY_S)
    r = ds.nuts(num_warmup=PARITY_W, num_samples=PARITY_S, num_chains=4, seed=50)
    posterior_parity(r.draws, o)


def _assert_recovery(results, true_params):  # nmixture.py:400-420
    assert np.allclose(results.samples["abundance"].mean(), true_params["abundance"].mean(), rtol=0.2)
    assert np.allclose([results.samples[k].mean() for k in [f"cov_state_{i}" for i in range(true_params["beta"].shape[1])]],
                       true_params["beta"].mean(axis=0), atol=0.5)
    assert np.allclose([results.samples[k].mean() for k in [f"cov_det_{i}" for i in range(true_params["alpha"].shape[1])]],
                       true_params["alpha"].mean(axis=0), atol=0.5)


def test_nmixture_like_reference():  # nmixture.py:372-420
    data, true_params = simulate_nmixture(**REF_TEST)
    max_abundance = int(np.nanmax(data["obs"]))
    results = fit(nmixture, **data, max_abundance=max_abundance, num_chains=1, num_samples=300, num_warmup=300, timeout=600)
    _assert_recovery(results, true_params)
    assert results.samples["abundance"].shape == (300, 1, 100, 1)
    assert results.samples["prob_detection"].shape == (300, 10, 1, 100, 1)


def test_nmixture_multi_season():  # nmixture.py:423-449
    data, true_params = simulate_nmixture(**REF_TEST, n_periods=3)
    max_abundance = int(np.nanmax(data["obs"]))
    results = fit(nmixture, **data, max_abundance=max_abundance, num_chains=1, num_samples=300, num_warmup=300, timeout=600)
    _assert_recovery(results, true_params)


def test_predict_nmixture_sites_and_distributions():
    data, _ = simulate_nmixture(**REF_TEST)
    K = int(np.nanmax(data["obs"])) + 5
    res = fit(nmixture, **data, max_abundance=K, num_chains=1, num_samples=200, num_warmup=200)
    preds = predict(nmixture, res.mcmc, **data, max_abundance=K, num_samples=None)
    assert set(preds) == {"abundance", "N_i", "prob_detection", "y"}
    assert preds["N_i"].shape == (200, 1, 100, 1) and preds["y"].shape == (200, 10, 1, 100, 1) and preds["y"].dtype == np.int32
    np.testing.assert_array_equal(preds["abundance"], res.samples["abundance"])
    Ni, y, p = preds["N_i"], preds["y"], preds["prob_detection"]
    assert Ni.max() <= K and (y <= Ni[:, None]).all() and (y >= 0).all()
    want = p * Ni[:, None]                                       # E[y | N, p]
    var = (p * (1 - p) * Ni[:, None]).sum()
    assert abs(y.sum() - want.sum()) < 5 * np.sqrt(var)
    # hand-made identical draws: N over draws is an i.i.d. truncated-Poisson sample
    n = 4000
    ds = OccuDataset(data["site_covs"][:8], data["obs_covs"][:8], np.full((1, 8, 1, 10), np.nan), model="nmixture", max_abundance=6)
    draws = np.tile(np.array([1.0, 0.0, 0.0, 0.0], np.float32), (n, 1))
    N6, _ = ds.predictive(draws, seed=2)
    from scipy.stats import poisson
    pmf = poisson.pmf(np.arange(7), np.e); pmf /= pmf.sum()
    cnt = np.bincount(N6[:, 0, 0], minlength=7) / n
    assert np.abs(cnt - pmf).max() < 5 * np.sqrt(0.25 / n)
