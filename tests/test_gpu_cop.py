"""Count occupancy kernel (occu_cop, MODEL 3) through the C-ABI (bl_dataset_create_cop) against the float64
oracle, plus the reference's own fit assertions (biolith/models/occu_cop.py:399-424)."""
import numpy as np
import pytest

import oracle
from biolith_amd.engine import OccuDataset
from biolith_amd.evaluation import diagnostics
from biolith_amd.models import occu_cop, simulate_cop
from biolith_amd.utils import fit, predict
from conftest import PARITY_S, PARITY_W, load_golden, posterior_parity

pytestmark = pytest.mark.gpu
# float32 per-term math; the Poisson terms y nu - d lambda are larger and less uniform than the Bernoulli ones
U_RTOL, G_RTOL = 2e-6, 2e-5


def _pair(name, mode, rate=1.0, priors=((0.0, 1.0), (0.0, 1.0))):
    g = load_golden(name)
    kw = dict(model="occu_cop", fp_mode=mode, prior_fp_rate=rate, session_duration=g["session_duration"])
    return (g, oracle.OracleData(g["site_covs"], g["obs_covs"], g["obs"], *priors, **kw),
            OccuDataset(g["site_covs"], g["obs_covs"], g["obs"], *priors, **kw))


@pytest.mark.parametrize("name,mode,rate", [("cop_default", None, 1.0), ("cop_default", "constant", 1.0), ("cop_missing", "constant", 2.0),
                                             ("cop_missing", "unoccupied", 1.0), ("cop_small_2x2", None, 1.0),
                                             ("cop_small_2x2", "unoccupied", 0.5), ("cop_small_2x2", "constant", 3.0)])
def test_cop_logp_grad_parity(name, mode, rate):
    _, od, ds = _pair(name, mode, rate, priors=((0.1, 1.5), (-0.2, 0.8)))
    assert ds.D == od.D
    th = np.random.default_rng(3).uniform(-1.2, 1.2, size=(5, od.D)).astype(np.float32).astype(np.float64)
    if mode:
        th[0, -1], th[1, -1] = -5.0, 0.7        # rates 0.0067 and 2.0
    Uo, Go = od.potential_grad(th)
    Ug, Gg = ds.logp_grad(th)
    assert np.all(np.isfinite(Ug)) and np.all(np.isfinite(Gg))
    assert np.max(np.abs(Ug - Uo) / np.abs(Uo)) <= U_RTOL, (Ug, Uo)
    assert np.max(np.abs(Gg - Go) / np.max(np.abs(Go), axis=1, keepdims=True)) <= G_RTOL, np.abs(Gg - Go).max(1)


@pytest.mark.parametrize("n_sites", [1, 2, 65, 385, 1031])
def test_cop_ragged_site_counts(n_sites):
    rng = np.random.default_rng(n_sites)
    X = rng.normal(size=(n_sites, 2)); W = rng.normal(size=(n_sites, 2, 3, 2)) * 0.5
    Y = rng.poisson(1.5, size=(1, n_sites, 2, 3)).astype(float)
    Y[rng.uniform(size=Y.shape) < 0.1] = np.nan
    Dur = rng.uniform(1.0, 9.0, size=(n_sites, 2, 3))
    kw = dict(model="occu_cop", fp_mode="constant", session_duration=Dur)
    od, ds = oracle.OracleData(X, W, Y, **kw), OccuDataset(X, W, Y, **kw)
    th = rng.uniform(-1.0, 1.0, size=(2, 7)).astype(np.float32).astype(np.float64)
    Uo, Go = od.potential_grad(th)
    Ug, Gg = ds.logp_grad(th)
    assert np.max(np.abs(Ug - Uo) / np.abs(Uo)) <= U_RTOL
    assert np.max(np.abs(Gg - Go)) <= G_RTOL * np.max(np.abs(Go))


@pytest.mark.parametrize("mode", [None, "constant"])
def test_cop_first_transitions_match_oracle(mode):
    _, od, ds = _pair("cop_small_2x2", mode)
    o = oracle.nuts_run(od, 0, 5, num_chains=2, seed=3)
    r = ds.nuts(num_warmup=0, num_samples=5, num_chains=2, seed=3)
    assert r.draws.shape == (2, 5, od.D)
    assert np.array_equal(o["num_steps"][:, :3], r.num_steps[:, :3]), (o["num_steps"], r.num_steps)
    assert np.allclose(o["draws"][:, 0], r.draws[:, 0], atol=2e-3)


@pytest.mark.parametrize("mode", [None, "constant"])
def test_cop_posterior_matches_oracle(mode):
    g, od, ds = _pair("cop_small_2x2", mode)
    # start at the generating parameters: with a false-positive rate the likelihood has a second mode
    # (swap detections and false positives), as for occu (test_gpu_fp.py)
    init = np.concatenate([g["beta"][0], g["alpha"][0]] + ([[np.log(0.12)]] if mode else []))
    init = np.tile(init, (4, 1))
    o = oracle.nuts_run(od, PARITY_W, PARITY_S, num_chains=4, seed=0, init=init)
    r = ds.nuts(num_warmup=PARITY_W, num_samples=PARITY_S, num_chains=4, seed=50, init_theta=init)
    posterior_parity(r.draws, o["draws"])


def test_occu_cop_like_reference():  # occu_cop.py:399-424 (simulate_cop passes false_positives_constant=True)
    data, true_params = simulate_cop(simulate_missing=True)
    results = fit(occu_cop, **data, timeout=600)
    assert np.allclose(results.samples["psi"].mean(), true_params["z"].mean(), atol=0.1)
    assert np.allclose([results.samples[k].mean() for k in [f"cov_state_{i}" for i in range(true_params["beta"].shape[1])]],
                       true_params["beta"].mean(axis=0), atol=0.5)
    assert np.allclose([results.samples[k].mean() for k in [f"cov_det_{i}" for i in range(true_params["alpha"].shape[1])]],
                       true_params["alpha"].mean(axis=0), atol=0.5)
    assert results.samples["rate_fp_constant"].shape == (5000,) and (results.samples["rate_fp_constant"] > 0).all()
    assert results.samples["psi"].shape == (5000, 1, 100, 1)
    assert results.samples["rate_detection"].shape == (5000, 52, 1, 100, 1)
    d = diagnostics(results.mcmc)
    assert set(d) >= {"n_eff_mean", "r_hat_max"} or isinstance(d, dict)
    preds = predict(occu_cop, results.mcmc, **data, num_samples=None)
    assert set(preds) == {"psi", "z", "rate_detection", "y"}
    assert preds["y"].shape == (5000, 52, 1, 100, 1) and preds["y"].dtype == np.int32 and preds["z"].shape == (5000, 1, 100, 1)
    np.testing.assert_array_equal(preds["psi"], results.samples["psi"])


def test_occu_cop_without_false_positives_and_deterministic_sites():
    data, true_params = simulate_cop(n_sites=150, n_site_covs=2, n_obs_covs=2, deployment_days_per_site=70, random_seed=2)
    data.pop("false_positives_constant")
    res = fit(occu_cop, **data, num_chains=2, num_samples=200, num_warmup=200)
    assert "rate_fp_constant" not in res.samples
    post = res.mcmc.get_samples()
    W = np.nan_to_num(np.asarray(data["obs_covs"], np.float32))
    nu = post["alpha"][:, 0, 0][:, None, None] + np.einsum("ijk,nk->nji", W[:, 0], post["alpha"][:, 0, 1:])
    np.testing.assert_allclose(res.samples["rate_detection"][:, :, 0, :, 0], np.exp(nu), rtol=2e-5)


def test_predictive_counts_follow_the_poisson_rates():
    # hand-made "posterior" with identical draws: the predictive sample over draws is i.i.d. from the model
    # (occu_cop.py:222-255); covers both Poisson samplers (rate < 10: inversion, else PTRS)
    N, J, n = 40, 3, 3000
    X = np.linspace(-1.0, 1.0, N, dtype=np.float32)[:, None]
    W = np.zeros((N, 1, J, 1), np.float32); W[:, 0, :, 0] = np.array([-1.0, 0.0, 1.5])
    Dur = np.tile(np.array([1.0, 4.0, 9.0], np.float32), (N, 1))[:, None, :]
    obs = np.full((1, N, 1, J), np.nan, np.float32)
    ds = OccuDataset(X, W, obs, model="occu_cop", fp_mode="constant", session_duration=Dur)
    beta, alpha, f = np.array([0.3, 1.0]), np.array([0.4, 0.9]), 0.25
    draws = np.tile(np.concatenate([beta, alpha, [np.log(f)]]).astype(np.float32), (n, 1))
    z, y = ds.predictive(draws, seed=5)
    assert z.dtype == np.int32 and y.dtype == np.int32 and z.shape == (n, 1, N) and y.shape == (n, J, 1, N)
    psi = 1 / (1 + np.exp(-(beta[0] + beta[1] * X[:, 0].astype(np.float64))))
    assert np.abs(z[:, 0].mean(0) - psi).max() < 5 * np.sqrt(0.25 / n)
    lam = np.exp(alpha[0] + alpha[1] * W[0, 0, :, 0].astype(np.float64))                       # (J,)
    rate = Dur[0, 0].astype(np.float64)[None, :, None] * (z[:, 0][:, None, :] * lam[None, :, None] + f)   # (n, J, N)
    yy = y[:, :, 0].astype(np.float64)
    assert abs(yy.sum() - rate.sum()) < 5 * np.sqrt(rate.sum())
    for j in range(J):                                                                         # mean and variance per visit type
        occ = z[:, 0] == 1
        r1 = Dur[0, 0, j] * (lam[j] + f)
        sample = yy[:, j][occ]
        assert abs(sample.mean() - r1) < 5 * np.sqrt(r1 / sample.size)
        assert abs(sample.var() / r1 - 1) < 0.1
        r0 = Dur[0, 0, j] * f
        assert abs(yy[:, j][~occ].mean() - r0) < 5 * np.sqrt(r0 / (~occ).sum())
