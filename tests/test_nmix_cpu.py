"""N-mixture model (biolith/models/nmixture.py) -- CPU side: the generator against fixtures made by importing the
reference's simulate_nmixture, the C oracle against the literal NumPy model and finite differences, the validator."""
import json
import os

import numpy as np
import pytest

import oracle
from biolith_amd.models import nmixture, simulate_nmixture
from conftest import GOLDEN, load_golden


@pytest.fixture(scope="module")
def nmix_index():
    with open(os.path.join(GOLDEN, "simulate_nmix_index.json")) as f:
        return json.load(f)


@pytest.mark.parametrize("name", ["nmix_default", "nmix_ref_test", "nmix_ref_test_3periods", "nmix_small_2x2", "nmix_site_re", "nmix_both_re"])
def test_simulate_nmixture_matches_reference(nmix_index, name, capsys):
    entry, g = nmix_index[name], load_golden(name)
    data, truth = simulate_nmixture(**entry["kwargs"])
    assert capsys.readouterr().out == entry["stdout"]
    for k in ("site_covs", "obs_covs", "obs"):
        assert np.array_equal(np.asarray(data[k], dtype=np.float64), g[k], equal_nan=True), k
    for k in ("N_i", "abundance", "beta", "alpha") + tuple(k for k in ("site_re_abu", "site_re_det", "obs_re") if k in g):
        assert np.array_equal(truth[k], g[k]), k
    assert data["coords"] is None and data["ell"] == entry["ell"]
    with pytest.raises(NotImplementedError):
        simulate_nmixture(spatial=True)


@pytest.mark.parametrize("name,K", [("nmix_ref_test", 9), ("nmix_ref_test", 30), ("nmix_ref_test_3periods", 19),
                                     ("nmix_small_2x2", 40), ("nmix_default", 100)])
def test_nmix_potential_equals_literal_model_and_fd(name, K):
    g = load_golden(name)
    od = oracle.OracleData(g["site_covs"], g["obs_covs"], g["obs"], (0.2, 1.5), (-0.1, 0.7), model="nmixture", max_abundance=K)
    rng = np.random.default_rng(3)
    for _ in range(2):
        th = rng.uniform(-1.0, 1.0, size=od.D)
        U, G = od.potential_grad(th)
        lit = oracle.literal_log_joint_nmix(th, g["site_covs"], g["obs_covs"], g["obs"], max_abundance=K,
                                            prior_beta=(0.2, 1.5), prior_alpha=(-0.1, 0.7))
        assert U == pytest.approx(-lit, rel=1e-12)
        h = 1e-6
        fd = np.array([(od.potential_grad(th + h * e)[0] - od.potential_grad(th - h * e)[0]) / (2 * h) for e in np.eye(od.D)])
        assert np.max(np.abs(fd - G)) <= 1e-7 * max(1.0, np.max(np.abs(G)))


@pytest.mark.parametrize("name,site,obs", [("nmix_ref_test", True, False), ("nmix_small_2x2", False, True), ("nmix_both_re", True, True),
                                           ("nmix_ref_test_3periods", True, True)])
def test_nmix_re_potential_equals_literal_model_and_fd(name, site, obs):
    """Random effects (nmixture.py:139-141, 166-172, 199-214): the oracle's closed form against the literal model and central differences."""
    g = load_golden(name)
    K = int(np.nanmax(g["obs"])) + 4
    kw = dict(site_random_effects=site, obs_random_effects=obs, prior_site_re_sd=0.8, prior_obs_re_sd=1.2)
    od = oracle.OracleData(g["site_covs"], g["obs_covs"], g["obs"], (0.2, 1.5), (-0.1, 0.7), model="nmixture", max_abundance=K, **kw)
    N, T, J = g["obs"].shape[1:]
    G = od.Ks + od.Ko + 2
    assert od.D == G + site * (1 + 2 * N) + obs * (1 + N * T * J)
    rng = np.random.default_rng(5)
    th = rng.uniform(-0.8, 0.8, size=od.D)
    U, Gr = od.potential_grad(th)
    lit = oracle.literal_log_joint_nmix(th, g["site_covs"], g["obs_covs"], g["obs"], max_abundance=K, prior_beta=(0.2, 1.5),
                                        prior_alpha=(-0.1, 0.7), **kw)
    assert U == pytest.approx(-lit, rel=1e-11)
    h = 1e-6
    idx = np.unique(np.concatenate([np.arange(min(G + 2, od.D)), rng.integers(0, od.D, size=12)]))
    fd = np.array([(od.potential_grad(th + h * e)[0] - od.potential_grad(th - h * e)[0]) / (2 * h) for e in np.eye(od.D)[idx]])
    assert np.max(np.abs(fd - Gr[idx])) <= 2e-6 * max(1.0, np.max(np.abs(Gr)))


def test_nmix_weights_are_not_renormalised():
    # the "N_i_trunc_norm" factor (nmixture.py:191) undoes the Categorical's normalisation: with everything masked the
    # (site, period) contributes log P(N <= K) under Poisson(lambda), not 0 as it would under occu_rn's renormalised prior
    from scipy.stats import poisson
    X = np.array([[0.3], [-0.4]]); W = np.zeros((2, 1, 3, 1)); Y = np.full((1, 2, 1, 3), np.nan)
    K = 3
    od = oracle.OracleData(X, W, Y, model="nmixture", max_abundance=K)
    th = np.array([0.5, 0.2, 0.1, -0.3])
    lam = np.exp(th[0] + th[1] * X[:, 0].astype(np.float32).astype(np.float64))
    prior = sum(-0.5 * t * t - 0.5 * np.log(2 * np.pi) for t in th)
    assert od.potential_grad(th)[0] == pytest.approx(-(np.log(poisson.cdf(K, lam)).sum() + prior), rel=1e-12)


def test_nmixture_validates_like_reference():
    g = load_golden("nmix_small_2x2")
    spec = nmixture(g["site_covs"], g["obs_covs"], obs=g["obs"], max_abundance=40)
    assert spec.model == "nmixture" and spec.extras["max_abundance"] == 40 and spec.shape["J"] == 6
    with pytest.raises(AssertionError, match="obs must have n_sites rows"):
        nmixture(g["site_covs"], g["obs_covs"], obs=g["obs"][:, :10])
    re = nmixture(g["site_covs"], g["obs_covs"], obs=g["obs"], max_abundance=40, site_random_effects=True)   # nmixture.py:139-141, 166, 199
    assert re.model == "nmixture" and re.extras["site_random_effects"] and not re.extras["obs_random_effects"]
    for bad in (dict(coords=np.zeros((60, 2))), dict(max_abundance=500)):
        with pytest.raises(NotImplementedError):
            nmixture(g["site_covs"], g["obs_covs"], obs=g["obs"], **bad)
