"""occu with a false-positive rate (biolith/models/occu.py:146-157, 229-241) through the C-ABI
(bl_dataset_create_fp) against the float64 oracle, plus the reference's own fit assertions
(occu.py:495-524) and the predictive sites."""
import numpy as np
import pytest

import oracle
from biolith_amd.distributions import Beta
from biolith_amd.engine import OccuDataset
from biolith_amd.models import occu, simulate
from biolith_amd.utils import fit, predict
from conftest import PARITY_S, PARITY_W, load_golden, posterior_parity

pytestmark = pytest.mark.gpu
U_RTOL, G_RTOL = 1e-6, 1e-5   # as for the plain occu kernel (test_gpu_logp.py)


def _pair(name, mode, prior=(2.0, 5.0), priors=((0.0, 1.0), (0.0, 1.0))):
    g = load_golden(name)
    kw = dict(model="occu_fp", fp_mode=mode, prior_fp=prior)
    return (g, oracle.OracleData(g["site_covs"], g["obs_covs"], g["obs"], *priors, **kw),
            OccuDataset(g["site_covs"], g["obs_covs"], g["obs"], *priors, **kw))


@pytest.mark.parametrize("name,mode,prior", [("fp_constant", "constant", (2.0, 5.0)), ("fp_constant", "unoccupied", (2.0, 5.0)),
                                             ("fp_unoccupied", "unoccupied", (1.0, 9.0)), ("fp_unoccupied", "constant", (3.0, 2.0)),
                                             ("missing_3periods", "constant", (2.0, 5.0)), ("small_3x3", "unoccupied", (0.5, 0.5))])
def test_fp_logp_grad_parity(name, mode, prior):
    _, od, ds = _pair(name, mode, prior, priors=((0.2, 1.5), (-0.1, 0.7)))
    assert ds.D == od.D
    th = np.random.default_rng(3).uniform(-2, 2, size=(5, od.D)).astype(np.float32).astype(np.float64)
    th[0, -1], th[1, -1] = -6.0, 4.0           # rates 0.0025 and 0.98
    Uo, Go = od.potential_grad(th)
    Ug, Gg = ds.logp_grad(th)
    assert np.all(np.isfinite(Ug)) and np.all(np.isfinite(Gg))
    assert np.max(np.abs(Ug - Uo) / np.abs(Uo)) <= U_RTOL, (Ug, Uo)
    assert np.max(np.abs(Gg - Go) / np.max(np.abs(Go), axis=1, keepdims=True)) <= G_RTOL, np.abs(Gg - Go).max(1)


@pytest.mark.parametrize("n_sites", [1, 2, 65, 513, 1031])
def test_fp_ragged_site_counts(n_sites):
    rng = np.random.default_rng(n_sites)
    X = rng.normal(size=(n_sites, 2)); W = rng.normal(size=(n_sites, 2, 3, 2))
    Y = (rng.uniform(size=(1, n_sites, 2, 3)) < 0.3) * 1.0
    Y[rng.uniform(size=Y.shape) < 0.1] = np.nan
    kw = dict(model="occu_fp", fp_mode="constant")
    od, ds = oracle.OracleData(X, W, Y, **kw), OccuDataset(X, W, Y, **kw)
    th = rng.uniform(-1.5, 1.5, size=(2, 7)).astype(np.float32).astype(np.float64)
    Uo, Go = od.potential_grad(th)
    Ug, Gg = ds.logp_grad(th)
    assert np.max(np.abs(Ug - Uo) / np.abs(Uo)) <= U_RTOL
    assert np.max(np.abs(Gg - Go)) <= G_RTOL * np.max(np.abs(Go))


def test_fp_limits():
    g = load_golden("small_3x3")
    X17 = np.zeros((300, 17), np.float32)
    with pytest.raises((NotImplementedError, RuntimeError, ValueError)):   # the engine's capacity: BL_MAX_COVS = 16 per side
        OccuDataset(X17, g["obs_covs"], g["obs"], model="occu_fp")
    ds5 = OccuDataset(np.zeros((300, 5), np.float32), g["obs_covs"], g["obs"], model="occu_fp")   # (round 1 stopped at 4)
    assert ds5.D == 5 + g["obs_covs"].shape[-1] + 3
    with pytest.raises(ValueError):
        OccuDataset(g["site_covs"], g["obs_covs"], g["obs"], model="occu_fp", prior_fp=(0.0, 1.0))


@pytest.mark.parametrize("mode", ["constant", "unoccupied"])
def test_fp_first_transitions_match_oracle(mode):
    _, od, ds = _pair("fp_unoccupied", mode)
    o = oracle.nuts_run(od, 0, 5, num_chains=2, seed=3)
    r = ds.nuts(num_warmup=0, num_samples=5, num_chains=2, seed=3)
    assert r.draws.shape == (2, 5, od.D)
    assert np.array_equal(o["num_steps"][:, :3], r.num_steps[:, :3]), (o["num_steps"], r.num_steps)
    assert np.allclose(o["draws"][:, 0], r.draws[:, 0], atol=2e-3)


@pytest.mark.parametrize("name,mode,rate", [("fp_constant", "constant", 0.1), ("fp_unoccupied", "unoccupied", 0.08)])
def test_fp_posterior_matches_oracle(name, mode, rate):
    # The false-positive likelihood is bimodal (swap "occupied, detected with p" with "unoccupied, false
    # positive with f": why the reference skips occu.py:527-560), and which mode a chain warms up in depends
    # on its start.  Both samplers therefore start every chain at the generating parameters, and the
    # comparison is of the mode around them.
    g, od, ds = _pair(name, mode)
    init = np.tile(np.concatenate([g["beta"][0], g["alpha"][0], [np.log(rate / (1 - rate))]]), (4, 1))
    o = oracle.nuts_run(od, PARITY_W, PARITY_S, num_chains=4, seed=0, init=init)
    r = ds.nuts(num_warmup=PARITY_W, num_samples=PARITY_S, num_chains=4, seed=50, init_theta=init)
    posterior_parity(r.draws, o["draws"])


def test_occu_fp_constant():  # occu.py:495-524
    prob_fp_constant = 0.1
    data, true_params = simulate(simulate_missing=True, prob_fp_constant=prob_fp_constant)
    results = fit(occu, **data, false_positives_constant=True, timeout=600)
    assert results.samples["prob_fp_constant"].shape == (5000,)
    assert np.allclose(results.samples["psi"].mean(), true_params["z"].mean(), atol=0.1)
    assert np.allclose(results.samples["prob_fp_constant"].mean(), prob_fp_constant, atol=0.1)
    assert np.allclose([results.samples[k].mean() for k in [f"cov_state_{i}" for i in range(true_params["beta"].shape[1])]],
                       true_params["beta"].mean(axis=0), atol=0.5)
    assert np.allclose([results.samples[k].mean() for k in [f"cov_det_{i}" for i in range(true_params["alpha"].shape[1])]],
                       true_params["alpha"].mean(axis=0), atol=0.5)


def test_occu_fp_unoccupied_recovers_the_rate():  # the reference skips its own version of this test (occu.py:527-560)
    data, true_params = simulate(n_sites=400, deployment_days_per_site=140, prob_fp_unoccupied=0.1, random_seed=2)
    results = fit(occu, **data, false_positives_unoccupied=True, prior_prob_fp_unoccupied=Beta(2, 5), num_chains=4)
    assert "prob_fp_unoccupied" in results.samples and "prob_fp_constant" not in results.samples
    assert np.allclose(results.samples["prob_fp_unoccupied"].mean(), 0.1, atol=0.05)
    assert np.allclose(results.samples["psi"].mean(), true_params["z"].mean(), atol=0.1)


def test_predict_with_false_positives():
    data, _ = simulate(n_sites=120, n_site_covs=1, n_obs_covs=1, deployment_days_per_site=70, prob_fp_constant=0.15, random_seed=4)
    res = fit(occu, **data, false_positives_constant=True, num_chains=2, num_samples=300, num_warmup=300)
    preds = predict(occu, res.mcmc, **data, false_positives_constant=True, num_samples=None)
    f = res.samples["prob_fp_constant"].reshape(-1, 1, 1, 1, 1)
    z, p = preds["z"][:, None], preds["prob_detection"]
    want = 1 - (1 - z * p) * (1 - f)
    np.testing.assert_allclose(preds["prob_detection_fp"], want, rtol=1e-5, atol=1e-6)
    y = preds["y"]
    assert abs(y.sum() - want.sum()) < 5 * np.sqrt((want * (1 - want)).sum())
    unocc = np.broadcast_to(z == 0, y.shape)
    assert y[unocc].any()                                       # false positives do appear at unoccupied sites
    assert abs(y[unocc].mean() - np.broadcast_to(f, y.shape)[unocc].mean()) < 0.01
    # fit().samples carries the site with z's enumeration axis (occu.py:229-235 under occu.py:208-210): [:, z]
    pfp = res.samples["prob_detection_fp"]
    assert pfp.shape == (600, 2) + p.shape[1:]
    pd = res.samples["prob_detection"]
    np.testing.assert_allclose(pfp[:, 0], np.broadcast_to(f, pd.shape), rtol=1e-6)                  # unoccupied: the rate alone
    np.testing.assert_allclose(pfp[:, 1], 1 - (1 - pd) * (1 - f), rtol=1e-5, atol=1e-6)             # occupied
