"""log_likelihood / lppd / waic (biolith/evaluation/log_likelihood.py, lppd.py, waic.py) on hand-made
predictive samples (CPU), and end to end after fit + predict as the reference's own tests do
(log_likelihood.py:99-128, lppd.py:98-123, waic.py:127-156) on the MI355X."""
import numpy as np
import pytest

from biolith_amd.evaluation import log_likelihood, log_likelihood_manual, lppd, lppd_manual, waic, waic_manual
from biolith_amd.models import occu, occu_rn, simulate


def _fake_predictive(rng, n, N, T, J, S=1):
    psi = rng.uniform(0.2, 0.9, size=(n, 1, N, S)).repeat(T, axis=1).astype(np.float32)
    p = rng.uniform(0.1, 0.8, size=(n, J, T, N, S)).astype(np.float32)
    z = (rng.uniform(size=psi.shape) < psi).astype(np.int32)
    return {"psi": psi, "prob_detection": p, "z": z, "prob_detection_fp": p * z[:, None],
            "y": (rng.uniform(size=p.shape) < p * z[:, None]).astype(np.int32)}


def _data(rng, N, T, J, missing=True):
    obs = (rng.uniform(size=(1, N, T, J)) < 0.3).astype(np.float32)
    site_covs = rng.normal(size=(N, 2)).astype(np.float32)
    obs_covs = rng.normal(size=(N, T, J, 2)).astype(np.float32)
    if missing:
        obs[0, 1, 0, 2] = np.nan
        obs_covs[3, 0, 1, 0] = np.nan
        site_covs[5, 1] = np.nan
    return dict(site_covs=site_covs, obs_covs=obs_covs, obs=obs)


def test_log_likelihood_is_the_clamped_bernoulli_log_prob():
    rng = np.random.default_rng(0)
    n, N, T, J = 7, 9, 2, 4
    ps, data = _fake_predictive(rng, n, N, T, J), _data(rng, N, T, J)
    ll = log_likelihood(occu, ps, **data)["y"]
    assert ll.shape == (n, J, T, N, 1) and ll.dtype == np.float32
    tiny, eps = np.finfo(np.float32).tiny, np.finfo(np.float32).eps
    for (q, j, t, i) in [(0, 0, 0, 0), (3, 2, 1, 4), (6, 3, 1, 8), (2, 1, 0, 7)]:
        pr = min(max(float(ps["prob_detection"][q, j, t, i, 0]) * int(ps["z"][q, t, i, 0]), tiny), 1 - eps)
        y = data["obs"][0, i, t, j]
        want = np.log(pr) if y == 1 else np.log1p(-pr)
        assert ll[q, j, t, i, 0] == pytest.approx(want, rel=1e-6, abs=1e-12)
    # unoccupied draw and a detection: log(finfo.tiny), NumPyro's clamp -- not -inf
    q, t, i = np.argwhere(ps["z"][..., 0] == 0)[0]
    j = 0
    data2 = {k: v.copy() for k, v in data.items()}
    data2["obs"][0, i, t, j] = 1.0
    data2["obs_covs"][i, t, j] = 0.0
    data2["site_covs"][i] = 0.0
    assert log_likelihood(occu, ps, **data2)["y"][q, j, t, i, 0] == pytest.approx(np.log(tiny), rel=1e-6)
    # masked entries contribute 0 (mask_missing_obs): missing obs, missing obs covariate, missing site covariate
    assert (ll[:, 2, 0, 1, 0] == 0).all() and (ll[:, 1, 0, 3, 0] == 0).all() and (ll[:, :, :, 5, 0] == 0).all()
    # the observed-site keys are dropped, not used (log_likelihood.py:42-44)
    ps_bad = dict(ps, y=np.zeros(3), s=np.zeros(3))
    np.testing.assert_array_equal(log_likelihood(occu, ps_bad, **data)["y"], ll)
    # without prob_detection_fp the probability is rebuilt from z and prob_detection
    ps_min = {k: v for k, v in ps.items() if k != "prob_detection_fp"}
    np.testing.assert_array_equal(log_likelihood(occu, ps_min, **data)["y"], ll)


def test_log_likelihood_rn_uses_the_abundance_draws():
    rng = np.random.default_rng(1)
    n, N, T, J = 5, 6, 1, 3
    r = rng.uniform(0.1, 0.6, size=(n, J, T, N, 1)).astype(np.float32)
    Ni = rng.integers(0, 5, size=(n, T, N, 1)).astype(np.int32)
    data = _data(rng, N, T, J, missing=False)
    ll = log_likelihood(occu_rn, {"prob_detection": r, "N_i": Ni, "abundance": np.ones((n, T, N, 1))}, **data)["y"]
    pr = np.clip(1 - (1 - r.astype(np.float64)) ** Ni[:, None], np.finfo(np.float32).tiny, 1 - np.finfo(np.float32).eps)
    y = data["obs"].transpose(3, 2, 1, 0)[None]
    np.testing.assert_allclose(ll, y * np.log(pr) + (1 - y) * np.log1p(-pr), rtol=2e-5, atol=1e-6)


def test_manual_forms_shapes_and_agreement_with_the_conditional_form():
    rng = np.random.default_rng(2)
    n, N, T, J = 4000, 6, 1, 3
    # a constant "posterior": z is then an i.i.d. Bernoulli(psi) sample and the conditional lppd
    # converges to the marginal one (what the reference asserts at rtol 1e-2, lppd.py:117-123)
    psi = np.broadcast_to(rng.uniform(0.3, 0.8, size=(1, T, N, 1)), (n, T, N, 1)).astype(np.float32)
    p = np.broadcast_to(rng.uniform(0.2, 0.7, size=(1, J, T, N, 1)), (n, J, T, N, 1)).astype(np.float32)
    z = (rng.uniform(size=psi.shape) < psi).astype(np.int32)
    ps = {"psi": psi, "prob_detection": p, "z": z}
    data = _data(rng, N, T, J)
    llm = log_likelihood_manual(ps, data)
    assert llm.shape == (n, 1, N, T, J)
    i, t, j = 2, 0, 1
    pj = float(p[0, j, t, i, 0]) * float(psi[0, t, i, 0])
    want = np.log(pj) if data["obs"][0, i, t, j] == 1 else np.log(1 - pj)
    assert llm[0, 0, i, t, j] == pytest.approx(want, rel=1e-6)
    a, b = lppd(occu, ps, **data), lppd_manual(ps, data)
    assert -np.inf < a < 0 and a == pytest.approx(b, rel=1e-2)
    w, wm = waic(occu, ps, **data), waic_manual(ps, data)
    for r in (w, wm):
        assert set(r) == {"waic", "p_waic", "lppd"} and all(np.isfinite(v) for v in r.values())
        assert r["waic"] == pytest.approx(-2 * (r["lppd"] - r["p_waic"]))
    assert w["lppd"] == a and wm["lppd"] == b and w["p_waic"] > 0
    assert wm["p_waic"] == pytest.approx(0.0, abs=1e-9)  # constant posterior: no variance in the marginal form
    # psi given without the period / species axes is broadcast like the reference does (log_likelihood.py:82-90)
    ps2 = dict(ps, psi=psi[:, 0, :, 0])
    np.testing.assert_allclose(log_likelihood_manual(ps2, data), llm, equal_nan=True)


@pytest.mark.gpu
def test_lppd_waic_after_fit_and_predict():  # lppd.py:98-123, waic.py:127-156, log_likelihood.py:99-128
    from biolith_amd.utils import fit, predict

    data, _ = simulate(simulate_missing=True)
    results = fit(occu, **data)
    ps = predict(occu, results.mcmc, **data, num_samples=None)
    a, b = lppd(occu, ps, **data), lppd_manual(ps, data)
    assert -np.inf < a < 0
    assert a == pytest.approx(b, rel=1e-2)
    for r in (waic(occu, ps, **data), waic_manual(ps, data)):
        assert all(np.isfinite(v) for v in r.values()) and r["p_waic"] > 0
    from scipy.special import logsumexp
    valid = (np.isfinite(data["obs"]) & np.isfinite(data["obs_covs"]).all(-1)[None]
             & np.isfinite(data["site_covs"]).all(-1)[None, :, None, None])
    ll = log_likelihood(occu, ps, **data)["y"].transpose(0, 4, 3, 2, 1)[:, valid]
    llm = log_likelihood_manual(ps, data)[:, valid]
    np.testing.assert_allclose(logsumexp(ll, 0) - np.log(len(ll)), logsumexp(llm, 0) - np.log(len(llm)), rtol=1e-1)


def test_deviance_residuals_and_ppc_on_hand_made_samples():  # deviance.py, residuals.py, posterior_predictive_check.py
    from biolith_amd.evaluation import deviance, deviance_manual, posterior_predictive_check, residuals

    rng = np.random.default_rng(4)
    n, N, T, J = 500, 12, 2, 3
    ps, data = _fake_predictive(rng, n, N, T, J), _data(rng, N, T, J)
    # deviance = -2 log mean_q exp(sum of valid log-likelihoods of draw q)
    ll = log_likelihood(occu, ps, **data)["y"].astype(np.float64)
    valid = (np.isfinite(data["obs"]) & np.isfinite(data["obs_covs"]).all(-1)[None]
             & np.isfinite(data["site_covs"]).all(-1)[None, :, None, None])
    per_draw = ll.transpose(0, 4, 3, 2, 1)[:, valid].sum(1)
    want = -2 * (np.log(np.mean(np.exp(per_draw - per_draw.max()))) + per_draw.max())
    assert deviance(occu, ps, **data) == pytest.approx(want, rel=1e-10)
    dm = deviance_manual(ps, data)
    assert 0 < dm < np.inf
    # residuals: o = z - psi everywhere; d = y - p where the draw is occupied, NaN elsewhere
    occ, det = residuals(ps, data["obs"])
    assert occ.shape == (n, T, N, 1) and det.shape == (n, 1, N, T, J)
    np.testing.assert_allclose(occ, ps["z"] - ps["psi"])
    q, t, i = np.argwhere(ps["z"][..., 0] == 1)[0]
    want_d = data["obs"][0, i, t, 1] - ps["prob_detection"][q, 1, t, i, 0]
    assert np.isnan(want_d) or det[q, 0, i, t, 1] == pytest.approx(want_d)
    q, t, i = np.argwhere(ps["z"][..., 0] == 0)[0]
    assert np.isnan(det[q, 0, i, t]).all()
    # posterior predictive check: a p-value in [0, 1]; replicated data drawn from the same (psi, p) as "observed" data
    # gives a p-value away from the extremes, grossly different observed data an extreme one
    psi = np.broadcast_to(rng.uniform(0.4, 0.8, size=(1, T, N, 1)), (n, T, N, 1))
    p = np.broadcast_to(rng.uniform(0.3, 0.6, size=(1, J, T, N, 1)), (n, J, T, N, 1))
    z = (rng.uniform(size=psi.shape) < psi).astype(np.int32)
    y = (rng.uniform(size=p.shape) < p * z[:, None]).astype(np.int32)
    good = y[0].transpose(3, 2, 1, 0).astype(float)          # one replicate plays the observed data (S, N, T, J)
    sam = {"psi": psi, "prob_detection": p, "z": z, "y": y}
    for group_by in ("site", "revisit"):
        for statistic in ("freeman-tukey", "chi-squared"):
            pv = posterior_predictive_check(sam, good, group_by=group_by, statistic=statistic)
            assert 0.02 < pv < 0.98, (group_by, statistic, pv)
    assert posterior_predictive_check(sam, np.ones_like(good)) < 0.01
    with pytest.raises(ValueError):
        posterior_predictive_check(sam, good, statistic="g-test")
    with pytest.raises(ValueError):
        posterior_predictive_check(sam, good, group_by="period")


def test_log_likelihood_of_the_count_models():
    """nmixture (Binomial(N_i, p), nmixture.py:206-220) and occu_cop (Poisson(dur (z rate + (1 - z) f_u + f_c)),
    occu_cop.py:222-255) have their own pointwise likelihoods instead of an opaque KeyError."""
    from scipy import stats

    from biolith_amd.models import nmixture, occu_cop

    rng = np.random.default_rng(3)
    n, N, T, J = 7, 6, 2, 3
    data = _data(rng, N, T, J)
    data["obs"] = np.where(np.isnan(data["obs"]), np.nan, rng.integers(0, 5, size=data["obs"].shape)).astype(np.float32)
    p = rng.uniform(0.1, 0.8, size=(n, J, T, N, 1)).astype(np.float32)
    n_i = rng.integers(0, 9, size=(n, T, N, 1)).astype(np.int32)
    ll = log_likelihood(nmixture, {"abundance": np.ones((n, T, N, 1), np.float32), "N_i": n_i, "prob_detection": p, "y": None}, **data)["y"]
    assert ll.shape == (n, J, T, N, 1)
    y = data["obs"].transpose((3, 2, 1, 0))
    want = stats.binom.logpmf(np.nan_to_num(y)[None], n_i[:, None], p)
    ok = np.isfinite(y) & np.isfinite(data["obs_covs"]).all(-1).transpose((2, 1, 0))[..., None] & np.isfinite(data["site_covs"]).all(-1)[None, None, :, None]
    assert np.allclose(ll[:, ok], want[:, ok], rtol=2e-5, atol=2e-5) and np.all(ll[:, ~ok] == 0)
    assert np.isneginf(ll[:, ok]).any()          # a count above N_i is impossible

    dur = rng.uniform(0.5, 3.0, size=(N, T, J)).astype(np.float32)
    z = rng.integers(0, 2, size=(n, T, N, 1)).astype(np.int32)
    lam = rng.uniform(0.1, 2.0, size=(n, J, T, N, 1)).astype(np.float32)
    f = rng.uniform(0.05, 0.3, size=n).astype(np.float32)
    ll = log_likelihood(occu_cop, {"psi": None, "z": z, "rate_detection": lam, "rate_fp_unoccupied": f, "y": None},
                        session_duration=dur, **data)["y"]
    mu = dur.transpose((2, 1, 0))[None, ..., None] * (z[:, None] * lam + (1 - z[:, None]) * f.reshape(-1, 1, 1, 1, 1))
    want = stats.poisson.logpmf(np.nan_to_num(y)[None], mu)
    assert np.allclose(ll[:, ok], want[:, ok], rtol=2e-5, atol=2e-5) and np.all(ll[:, ~ok] == 0)
    # without a false-positive rate an unoccupied site cannot produce a count
    ll0 = log_likelihood(occu_cop, {"z": z, "rate_detection": lam, "y": None}, session_duration=dur, **data)["y"]
    bad = (z[:, None] == 0) & (np.nan_to_num(y)[None] > 0) & ok[None]
    assert np.all(np.isneginf(ll0[bad])) and np.all(ll0[(z[:, None] == 0) & (np.nan_to_num(y)[None] == 0) & ok[None]] == 0)
