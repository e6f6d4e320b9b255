"""Count occupancy model (biolith/models/occu_cop.py) -- CPU side: the generator against fixtures made by
importing the reference's simulate_cop, the C oracle against the literal NumPy model and finite
differences, and the validator."""
import json
import os

import numpy as np
import pytest

import oracle
from biolith_amd.distributions import Exponential, Normal
from biolith_amd.models import occu_cop, simulate_cop
from conftest import GOLDEN, load_golden, quiet_simulate


@pytest.fixture(scope="module")
def cop_index():
    with open(os.path.join(GOLDEN, "simulate_cop_index.json")) as f:
        return json.load(f)


@pytest.mark.parametrize("name", ["cop_default", "cop_missing", "cop_small_2x2"])
def test_simulate_cop_matches_reference(cop_index, name, capsys):
    entry, g = cop_index[name], load_golden(name)
    data, truth = simulate_cop(**entry["kwargs"])
    assert capsys.readouterr().out == entry["stdout"]
    for k in ("site_covs", "obs_covs", "obs", "session_duration"):
        assert np.array_equal(np.asarray(data[k], dtype=np.float64), g[k], equal_nan=True), k
    assert np.array_equal(truth["z"], g["z"]) and np.array_equal(truth["beta"], g["beta"]) and np.array_equal(truth["alpha"], g["alpha"])
    assert data["false_positives_constant"] is True and data["coords"] is None and data["ell"] == entry["ell"]
    with pytest.raises(NotImplementedError):
        simulate_cop(spatial=True)


@pytest.mark.parametrize("name,mode,rate", [("cop_default", None, 1.0), ("cop_missing", "constant", 1.0),
                                             ("cop_small_2x2", "unoccupied", 2.5), ("cop_small_2x2", "constant", 0.5)])
def test_cop_potential_equals_literal_model_and_fd(name, mode, rate):
    g = load_golden(name)
    kw = dict(fp_mode=mode, prior_fp_rate=rate)
    od = oracle.OracleData(g["site_covs"], g["obs_covs"], g["obs"], model="occu_cop", session_duration=g["session_duration"], **kw)
    assert od.D == g["site_covs"].shape[1] + g["obs_covs"].shape[3] + 2 + (1 if mode else 0)
    rng = np.random.default_rng(3)
    for _ in range(3):
        th = rng.uniform(-1.0, 1.0, size=od.D)
        U, G = od.potential_grad(th)
        lit = oracle.literal_log_joint_cop(th, g["site_covs"], g["obs_covs"], g["obs"], g["session_duration"], **kw)
        assert U == pytest.approx(-lit, rel=1e-12)
        h = 1e-6
        fd = np.array([(od.potential_grad(th + h * e)[0] - od.potential_grad(th - h * e)[0]) / (2 * h) for e in np.eye(od.D)])
        assert np.max(np.abs(fd - G)) <= 1e-7 * max(1.0, np.max(np.abs(G)))


@pytest.mark.parametrize("name,site,obs,mode", [("cop_small_2x2", True, False, None), ("cop_small_2x2", False, True, "unoccupied"),
                                                ("cop_missing", True, True, None), ("cop_missing", True, True, "constant")])
def test_cop_re_potential_equals_literal_model_and_fd(name, site, obs, mode):
    """Random effects (occu_cop.py:183-186, 204-210, 229-243), with and without a false-positive rate: the oracle against the literal
    model and central differences; theta = [beta, alpha, (phi = log rate_fp), (log sds), (effects)]."""
    g = load_golden(name)
    kw = dict(site_random_effects=site, obs_random_effects=obs, prior_site_re_sd=0.8, prior_obs_re_sd=1.2, fp_mode=mode, prior_fp_rate=2.0)
    od = oracle.OracleData(g["site_covs"], g["obs_covs"], g["obs"], (0.2, 1.5), (-0.1, 0.7), model="occu_cop",
                           session_duration=g["session_duration"], **kw)
    N, T, J = g["obs"].shape[1:]
    G = od.Ks + od.Ko + 2
    assert od.D == G + (mode is not None) + site * (1 + 2 * N) + obs * (1 + N * T * J)
    rng = np.random.default_rng(5)
    th = rng.uniform(-0.8, 0.8, size=od.D)
    U, grad = od.potential_grad(th)
    lit = oracle.literal_log_joint_cop(th, g["site_covs"], g["obs_covs"], g["obs"], g["session_duration"], prior_beta=(0.2, 1.5),
                                       prior_alpha=(-0.1, 0.7), **kw)
    assert abs(U + lit) <= 1e-10 * abs(U)
    h = 1e-6
    idx = np.unique(np.concatenate([np.arange(min(G + 2, od.D)), rng.integers(0, od.D, size=12)]))
    fd = np.array([(od.potential_grad(th + h * e)[0] - od.potential_grad(th - h * e)[0]) / (2 * h) for e in np.eye(od.D)[idx]])
    assert np.max(np.abs(fd - grad[idx])) <= 1e-6 * max(1.0, np.max(np.abs(grad)))


def test_cop_detection_at_an_unoccupied_site_is_impossible_without_false_positives():
    # Poisson(0) puts no mass on y > 0 (numpyro: xlogy(y, 0) = -inf): a site-period with any count is occupied for sure
    g = load_golden("cop_default")
    od = oracle.OracleData(g["site_covs"], g["obs_covs"], g["obs"], model="occu_cop", session_duration=g["session_duration"], fp_mode=None)
    th = np.array([-30.0, 0.0, 0.3, 0.1])   # psi ~ 1e-13
    U, G = od.potential_grad(th)
    assert np.isfinite(U) and np.all(np.isfinite(G))
    n_detected_sites = int((np.nansum(g["obs"][0], axis=(1, 2)) > 0).sum())
    assert U > 30.0 * n_detected_sites          # every detected site pays log psi ~ -30


def test_occu_cop_validates_like_reference():
    g = load_golden("cop_small_2x2")
    kw = dict(site_covs=g["site_covs"], obs_covs=g["obs_covs"], obs=g["obs"], session_duration=g["session_duration"])
    spec = occu_cop(**kw)
    assert spec.model == "occu_cop" and spec.shape == dict(S=1, N=80, T=2, J=6, Ks=2, Ko=2) and spec.extras["fp_mode"] is None
    fp = occu_cop(**kw, false_positives_constant=True, prior_rate_fp_constant=Exponential(2.0))
    assert fp.extras["fp_mode"] == "constant" and fp.extras["prior_fp_rate"] == 2.0
    assert occu_cop(**kw, false_positives_unoccupied=True).extras["fp_mode"] == "unoccupied"
    no_dur = occu_cop(g["site_covs"], g["obs_covs"], obs=g["obs"])       # occu_cop.py:146-148
    assert np.array_equal(no_dur.extras["session_duration"], np.ones((80, 2, 6), np.float32))
    with pytest.raises(AssertionError, match="cannot both be True"):
        occu_cop(**kw, false_positives_constant=True, false_positives_unoccupied=True)
    with pytest.raises(AssertionError, match="session_duration must have n_sites rows"):
        occu_cop(g["site_covs"], g["obs_covs"], obs=g["obs"], session_duration=g["session_duration"][:10])
    with pytest.raises(NotImplementedError, match="Exponential"):
        occu_cop(**kw, false_positives_constant=True, prior_rate_fp_constant=Normal())
    with pytest.raises(NotImplementedError):
        occu_cop(**kw, coords=np.zeros((80, 2)))
    re = occu_cop(**kw, site_random_effects=True)   # occu_cop.py:183-186
    assert re.extras["site_random_effects"] and not re.extras["obs_random_effects"] and re.extras["prior_site_re_sd"] == 1.0
    both = occu_cop(**kw, obs_random_effects=True, false_positives_unoccupied=True)   # (what the reference's own random-effects tests fit)
    assert both.extras["fp_mode"] == "unoccupied" and both.extras["obs_random_effects"]
    with pytest.raises(NotImplementedError, match="shared across species"):
        occu_cop(g["site_covs"], g["obs_covs"], obs=np.concatenate([g["obs"], g["obs"]]), session_duration=g["session_duration"],
                 false_positives_constant=True)
