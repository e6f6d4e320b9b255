"""Pin what can be pinned of the CPU oracle (oracle/occu_oracle.c) without the reference's
numpyro/jax half: closed form vs a literal z-enumerating restatement, analytic vs numerical
gradient, the survey's float64 anchors, NumPyro's adaptation-schedule known answers, and the
reference's own statistical-recovery tolerances (biolith/models/occu.py:440-456, 473-475)."""
import numpy as np
import pytest
from scipy.optimize import minimize

import oracle
from conftest import load_golden


def _data(name):
    g = load_golden(name)
    return g, oracle.OracleData(g["site_covs"], g["obs_covs"], g["obs"])


@pytest.mark.parametrize("name", ["default", "missing", "missing_3periods", "small_3x3", "seed7_2x1"])
def test_closed_form_equals_literal_enumeration(name):
    g, od = _data(name)
    rng = np.random.default_rng(3)
    for _ in range(4):
        th = rng.uniform(-1.5, 1.5, od.D)
        U, _ = od.potential_grad(th)
        lit = oracle.literal_log_joint(th, g["site_covs"], g["obs_covs"], g["obs"][0])
        assert abs(U + lit) <= 1e-10 * abs(U)
        # numpyro's prob clamp in the z=1 branch only matters for |nu| > 15.9: invisible here
        lit_c = oracle.literal_log_joint(th, g["site_covs"], g["obs_covs"], g["obs"][0], clamp_z1=True)
        assert abs(U + lit_c) <= 1e-6 * abs(U)


@pytest.mark.parametrize("name", ["default", "missing_3periods", "small_3x3"])
def test_gradient_matches_central_differences(name):
    _, od = _data(name)
    th = np.random.default_rng(4).uniform(-1, 1, od.D)
    _, g = od.potential_grad(th)
    h = 1e-5
    fd = np.array([(od.potential_grad(th + h * e)[0] - od.potential_grad(th - h * e)[0]) / (2 * h)
                   for e in np.eye(od.D)])
    assert np.max(np.abs(fd - g)) <= 1e-5 * max(1.0, np.max(np.abs(g)))


# SURVEY.md Appendix C "sanity anchors" (surveyor's independent float64 code; not reference output)
ANCHORS = {
    "default": (3268.686981, [-0.1608, 0.0870, 0.6347, 0.1512], 1610.7893, [0.1975, 0.2023, 0.0431, 0.0431]),
    "missing": (2391.248163, [-0.1943, 0.1619, 0.6282, 0.1233], 1184.3520, [0.2031, 0.2079, 0.0505, 0.0506]),
    "missing_3periods": (7291.697317, [-0.0931, 0.0182, 0.6608, 0.1020], 3552.3895, [0.1193, 0.1209, 0.0292, 0.0294]),
}


@pytest.mark.parametrize("name", list(ANCHORS))
def test_survey_anchors(name):
    U_ref, map_ref, Umap_ref, sd_ref = ANCHORS[name]
    _, od = _data(name)
    th = np.random.default_rng(1).uniform(-2, 2, od.D)
    U, _ = od.potential_grad(th)
    assert abs(U - U_ref) <= 2e-6 * U_ref
    res = minimize(lambda t: od.potential_grad(t), np.zeros(od.D), jac=True, method="BFGS", options=dict(gtol=1e-8))
    assert np.allclose(res.x, map_ref, atol=2e-3)
    assert abs(res.fun - Umap_ref) <= 1e-3
    # Laplace sd from a finite-difference Hessian of the analytic gradient
    h = 1e-5
    H = np.array([(od.potential_grad(res.x + h * e)[1] - od.potential_grad(res.x - h * e)[1]) / (2 * h)
                  for e in np.eye(od.D)])
    sd = np.sqrt(np.diag(np.linalg.inv(0.5 * (H + H.T))))
    assert np.allclose(sd, sd_ref, rtol=0.02)


def test_masking_rules():
    """NaN obs covariate masks that visit; NaN site covariate masks the whole site (occu.py:136-142)."""
    rng = np.random.default_rng(0)
    N, T, J = 6, 2, 3
    X = rng.normal(size=(N, 2)); W = rng.normal(size=(N, T, J, 1)); Y = (rng.uniform(size=(N, T, J)) < 0.4) * 1.0
    th = rng.normal(size=5)
    base = oracle.OracleData(X, W, Y).potential_grad(th)[0]
    W2 = W.copy(); W2[1, 0, 2, 0] = np.nan
    Y2 = Y.copy(); Y2[1, 0, 2] = np.nan
    assert oracle.OracleData(X, W2, Y).potential_grad(th)[0] == pytest.approx(
        oracle.OracleData(X, np.nan_to_num(W2), Y2).potential_grad(th)[0], rel=1e-14)
    X3 = X.copy(); X3[4, 1] = np.nan
    Y3 = Y.copy(); Y3[4] = np.nan
    assert oracle.OracleData(X3, W, Y).potential_grad(th)[0] == pytest.approx(
        oracle.OracleData(np.nan_to_num(X3), W, Y3).potential_grad(th)[0], rel=1e-14)
    assert base != oracle.OracleData(X3, W, Y).potential_grad(th)[0]
    # a fully masked dataset leaves only the prior
    Yn = np.full_like(Y, np.nan)
    prior = 0.5 * np.sum(th ** 2) + 5 * 0.9189385332046727
    assert oracle.OracleData(X, W, Yn).potential_grad(th)[0] == pytest.approx(prior, rel=1e-12)


def test_detection_in_unoccupied_branch_uses_float32_tiny():
    """One site, psi ~ 0, one detection: l = logaddexp(log psi + log p, log(1-psi) + log tiny)."""
    X = np.zeros((1, 1)); W = np.zeros((1, 1, 1, 1)); Y = np.ones((1, 1, 1))
    th = np.array([-200.0, 0.0, -200.0, 0.0])  # psi = p = e^-200: z=1 branch ~ -400, z=0 branch ~ -87.3
    U, _ = oracle.OracleData(X, W, Y).potential_grad(th)
    prior = 0.5 * np.sum(th ** 2) + 4 * 0.9189385332046727
    assert U - prior == pytest.approx(87.33654475, rel=1e-9)


def test_adaptation_schedule_known_answers():
    """SURVEY.md App. B.3 (numpyro build_adaptation_schedule)."""
    assert oracle.adaptation_schedule(1000) == [(0, 74), (75, 99), (100, 149), (150, 249), (250, 449), (450, 949), (950, 999)]
    assert oracle.adaptation_schedule(100) == [(0, 14), (15, 89), (90, 99)]
    assert oracle.adaptation_schedule(300) == [(0, 74), (75, 99), (100, 149), (150, 249), (250, 299)]
    assert oracle.adaptation_schedule(10) == [(0, 9)]


def test_oracle_nuts_is_deterministic_and_chains_differ():
    _, od = _data("seed7_2x1")
    a = oracle.nuts_run(od, 40, 30, num_chains=2, seed=11)
    b = oracle.nuts_run(od, 40, 30, num_chains=2, seed=11)
    assert np.array_equal(a["draws"], b["draws"])
    assert not np.allclose(a["draws"][0], a["draws"][1])
    c = oracle.nuts_run(od, 40, 30, num_chains=1, seed=11, chain_offset=1)
    assert np.array_equal(c["draws"][0], a["draws"][1])  # chain_offset selects the same stream
    assert (a["num_steps"] >= 1).all() and (a["num_steps"] <= 1023).all()


def test_oracle_recovers_truth_like_reference_test_occu():
    """biolith/models/occu.py:433-456 (test_occu) tolerances, on the same simulated data."""
    g, od = _data("missing")
    r = oracle.nuts_run(od, 500, 500, num_chains=4, seed=0)
    draws = r["draws"].reshape(-1, od.D)
    X = np.nan_to_num(g["site_covs"].astype(np.float32).astype(np.float64))
    psi = 1 / (1 + np.exp(-(draws[:, :1] + draws[:, 1:2] @ X.T)))
    assert abs(psi.mean() - g["z"].mean()) <= 0.1
    assert np.allclose(draws[:, :2].mean(0), g["beta"].mean(0), atol=0.5)
    assert np.allclose(draws[:, 2:].mean(0), g["alpha"].mean(0), atol=0.5)
    assert oracle.split_gelman_rubin(r["draws"]).max() < 1.02
    assert r["diverging"].sum() == 0
    # posterior vs Laplace anchor: mean within 4 MCSE (+ small skew allowance), sd within 15 %
    _, map_ref, _, sd_ref = ANCHORS["missing"]
    ess = oracle.effective_sample_size(r["draws"])
    mcse = draws.std(0) / np.sqrt(ess)
    assert np.all(np.abs(draws.mean(0) - map_ref) <= 4 * mcse + 0.15 * np.array(sd_ref))
    assert np.all(np.abs(draws.std(0) / sd_ref - 1) < 0.15)


def test_oracle_multi_season_recovery():
    """biolith/models/occu.py:459-475 (test_occu_multi_season): mean psi within 0.15."""
    g, od = _data("missing_3periods")
    r = oracle.nuts_run(od, 300, 300, num_chains=1, seed=0)
    draws = r["draws"].reshape(-1, od.D)
    X = np.nan_to_num(g["site_covs"].astype(np.float32).astype(np.float64))
    psi = 1 / (1 + np.exp(-(draws[:, :1] + draws[:, 1:2] @ X.T)))
    assert abs(psi.mean() - g["z"].mean()) <= 0.15
