"""predict() (biolith/utils/predict.py:9-94) on the MI355X: the posterior-predictive sites the
reference's Predictive call returns, checked for shape/dtype, exact structural identities
(y = 0 wherever the latent state is 0; deterministic sites equal the closed form) and, since JAX's
threefry draws are not reproduced, the sampling distributions against their exact means."""
import numpy as np
import pytest

from biolith_amd.engine import OccuDataset
from biolith_amd.models import occu, occu_rn, simulate, simulate_rn
from biolith_amd.utils import fit, predict

pytestmark = pytest.mark.gpu


def _sigmoid(x):
    return 1.0 / (1.0 + np.exp(-x))


def test_predict_occu_sites_shapes_and_identities():
    data, _ = simulate(n_sites=200, n_site_covs=2, n_obs_covs=2, deployment_days_per_site=35, random_seed=4)
    res = fit(occu, **data, num_chains=2, num_samples=150, num_warmup=150)
    with pytest.warns(UserWarning, match="num_samples"):
        preds = predict(occu, res.mcmc, **data, num_samples=5)  # Predictive: one draw per posterior draw
    n, N, T, J = 300, 200, 1, 5
    assert set(preds) == {"psi", "z", "prob_detection", "prob_detection_fp", "y"}
    assert preds["psi"].shape == (n, T, N, 1) and preds["z"].shape == (n, T, N, 1)
    assert preds["prob_detection"].shape == preds["y"].shape == preds["prob_detection_fp"].shape == (n, J, T, N, 1)
    assert preds["z"].dtype == np.int32 and preds["y"].dtype == np.int32
    # deterministic sites: the closed form on the posterior draws (occu.py:207,221-228)
    post = res.mcmc.get_samples()
    X = np.asarray(data["site_covs"], np.float64)
    eta = post["beta"][:, 0, :1] + post["beta"][:, 0, 1:].astype(np.float64) @ X.T
    np.testing.assert_allclose(preds["psi"][:, 0, :, 0], _sigmoid(eta), rtol=2e-5, atol=2e-6)
    np.testing.assert_allclose(preds["psi"], res.samples["psi"], rtol=0, atol=0)  # same covariates -> same site
    z, y, p = preds["z"], preds["y"], preds["prob_detection"]
    assert set(np.unique(z)) <= {0, 1} and set(np.unique(y)) <= {0, 1}
    assert not y[np.broadcast_to(z[:, None] == 0, y.shape)].any()          # unoccupied -> never detected
    np.testing.assert_array_equal(preds["prob_detection_fp"], p * z[:, None])
    # sampling distributions: sum of independent Bernoullis vs its exact mean, 5 sigma
    m, s = preds["psi"].sum(), np.sqrt((preds["psi"] * (1 - preds["psi"])).sum())
    assert abs(z.sum() - m) < 5 * s
    pz = p * z[:, None]
    assert abs(y.sum() - pz.sum()) < 5 * np.sqrt((pz * (1 - pz)).sum())
    # per-site occupancy frequency follows psi (not just the grand mean)
    freq, want = z[:, 0, :, 0].mean(0), preds["psi"][:, 0, :, 0].mean(0)
    assert np.abs(freq - want).max() < 5 * np.sqrt(0.25 / n)


def test_predict_is_seeded_and_chunk_independent():
    data, _ = simulate(n_sites=64, n_site_covs=1, n_obs_covs=1, deployment_days_per_site=21, random_seed=1)
    res = fit(occu, **data, num_chains=1, num_samples=40, num_warmup=40)
    kw = dict(num_samples=40)
    a = predict(occu, res.mcmc, **data, random_seed=3, **kw)
    b = predict(occu, res.mcmc, **data, random_seed=3, **kw)
    c = predict(occu, res.mcmc, **data, random_seed=4, **kw)
    assert np.array_equal(a["y"], b["y"]) and np.array_equal(a["z"], b["z"])
    assert not np.array_equal(a["z"], c["z"])
    # the draw for (posterior draw, period, site) does not depend on which other draws are in the call
    ds = OccuDataset(data["site_covs"], data["obs_covs"], data["obs"])
    d = res.mcmc.result.draws.reshape(-1, ds.D)
    z_all, y_all = ds.predictive(d, seed=3)
    z_head, y_head = ds.predictive(d[:7], seed=3)
    assert np.array_equal(z_all[:7], z_head) and np.array_equal(y_all[:7], y_head)
    assert np.array_equal(z_all[..., None].astype(np.int32), a["z"])


def test_predict_on_new_sites():  # grid_search.py:85-92 style: predict for held-out sites
    data, _ = simulate(n_sites=120, n_site_covs=2, n_obs_covs=2, deployment_days_per_site=28, random_seed=2)
    res = fit(occu, **data, num_chains=1, num_samples=50, num_warmup=50)
    rng = np.random.default_rng(0)
    new_site, new_obs = rng.normal(size=(33, 2)), rng.normal(size=(33, 1, 6, 2))
    preds = predict(occu, res.mcmc, site_covs=new_site, obs_covs=new_obs, num_samples=50)
    assert preds["psi"].shape == (50, 1, 33, 1) and preds["y"].shape == (50, 6, 1, 33, 1)
    post = res.mcmc.get_samples()
    nu = post["alpha"][:, 0, 0][:, None, None] + np.einsum("ijk,nk->nji", new_obs[:, 0].astype(np.float32), post["alpha"][:, 0, 1:])
    np.testing.assert_allclose(preds["prob_detection"][:, :, 0, :, 0], _sigmoid(nu), rtol=2e-5, atol=2e-6)
    with pytest.raises(ValueError):
        predict(occu, res.mcmc, site_covs=rng.normal(size=(33, 3)), obs_covs=new_obs, num_samples=50)
    # infer_discrete=True (predict.py:18, 71): obs is withheld from the model call (predict.py:78-80), so (z, y) are drawn from their
    # joint distribution given the posterior draw -- the sites and the distribution of the default path (same seed: the same sample)
    import warnings

    with warnings.catch_warnings():      # silent, as the reference hands the flag to Predictive (predict.py:67-72): pipelines with -W error
        warnings.simplefilter("error")
        disc = predict(occu, res.mcmc, site_covs=new_site, obs_covs=new_obs, num_samples=50, infer_discrete=True)
    assert set(disc.keys()) == set(preds.keys())
    assert np.array_equal(disc["z"], preds["z"]) and np.array_equal(disc["y"], preds["y"])


def test_predict_rn_abundance_and_detection_distributions():
    # hand-made "posterior": every draw has the same coefficients, so the predictive sample over draws
    # is an i.i.d. sample from the model at those coefficients (occu_rn.py:192-221)
    N, J, n, K = 48, 4, 4000, 30
    X = np.linspace(-1.0, 1.0, N, dtype=np.float32)[:, None]
    W = np.zeros((N, 1, J, 1), np.float32)
    W[:, 0, :, 0] = np.linspace(-1, 1, J)
    obs = np.full((1, N, 1, J), np.nan, np.float32)
    ds = OccuDataset(X, W, obs, model="occu_rn", max_abundance=K)
    beta, alpha = np.array([1.2, 0.8]), np.array([-1.0, 0.5])
    draws = np.tile(np.concatenate([beta, alpha]).astype(np.float32), (n, 1))
    Ni, y = ds.predictive(draws, seed=11)
    assert Ni.shape == (n, 1, N) and y.shape == (n, J, 1, N) and Ni.max() <= K
    lam = np.exp(beta[0] + beta[1] * X[:, 0].astype(np.float64))
    k = np.arange(K + 1)
    from scipy.special import gammaln
    logp = k[None] * np.log(lam[:, None]) - lam[:, None] - gammaln(k + 1)[None]
    pmf = np.exp(logp)
    pmf /= pmf.sum(1, keepdims=True)                      # Categorical(logits) renormalises (distributions.py:36-40)
    mean, var = (pmf * k).sum(1), (pmf * k ** 2).sum(1) - (pmf * k).sum(1) ** 2
    assert np.abs(Ni[:, 0].mean(0) - mean).max() < 5 * np.sqrt(var.max() / n)
    # whole pmf at one site (chi-square-ish: each cell within 5 sigma)
    cnt = np.bincount(Ni[:, 0, N // 2], minlength=K + 1) / n
    assert np.abs(cnt - pmf[N // 2]).max() < 5 * np.sqrt(0.25 / n)
    r = _sigmoid(alpha[0] + alpha[1] * W[0, 0, :, 0].astype(np.float64))          # (J,)
    pdet = 1.0 - (1.0 - r)[None, :, None] ** Ni[:, 0][:, None, :]                 # (n, J, N)
    assert abs(y[:, :, 0].sum() - pdet.sum()) < 5 * np.sqrt((pdet * (1 - pdet)).sum())
    assert not y[:, :, 0][np.broadcast_to(Ni[:, 0][:, None] == 0, (n, J, N))].any()


def test_predict_occu_rn_end_to_end():
    data, _ = simulate_rn(n_sites=60, n_site_covs=2, n_obs_covs=2, deployment_days_per_site=42, random_seed=3)
    res = fit(occu_rn, **data, num_chains=1, num_samples=40, num_warmup=40)
    preds = predict(occu_rn, res.mcmc, **data, num_samples=40)
    assert set(preds) == {"abundance", "N_i", "prob_detection", "y"}
    assert preds["N_i"].shape == preds["abundance"].shape == (40, 1, 60, 1)
    assert preds["y"].shape == preds["prob_detection"].shape
    np.testing.assert_array_equal(preds["abundance"], res.samples["abundance"])


def test_predict_multi_species():
    data, _ = simulate(n_species=2, n_sites=50, random_seed=6)
    res = fit(occu, **data, num_chains=1, num_samples=30, num_warmup=30)
    preds = predict(occu, res.mcmc, **data, num_samples=30)
    assert preds["z"].shape == (30, 1, 50, 2) and preds["y"].shape[-1] == 2
    assert not np.array_equal(preds["z"][..., 0], preds["z"][..., 1])
    np.testing.assert_array_equal(preds["psi"], res.samples["psi"])
