"""occu_cop with random effects (biolith/models/occu_cop.py:183-186, 204-210, 229-243): site_re_occ joins the occupancy predictor,
site_re_det and obs_re the log detection rate.  theta = [beta, alpha, (log sds), (effects)].  The kernels (re_kernel.hpp, kind 6)
through the C-ABI (bl_dataset_create_cop_re) against the float64 oracle: potential + gradient over every coordinate, the first trees
on shared streams (one and several workgroups per chain), the posterior, predict, and the reference's own three fit tests
(occu_cop.py:473-545), which sample the effects together with a false-positive rate (kind 7)."""
import contextlib
import io

import numpy as np
import pytest

import oracle
from biolith_amd.engine import OccuDataset
from biolith_amd.models import occu_cop, simulate_cop
from biolith_amd.utils import fit, predict
from conftest import load_golden, posterior_parity

pytestmark = pytest.mark.gpu

CASES = [("cop_small_2x2", True, False, None), ("cop_small_2x2", False, True, "unoccupied"), ("cop_missing", True, True, "constant"),
         ("cop_default", True, False, "constant"), ("cop_missing", True, True, None)]


def _pair(name, site, obs, mode):
    g = load_golden(name)
    kw = dict(model="occu_cop", session_duration=g["session_duration"], fp_mode=mode, prior_fp_rate=2.0, site_random_effects=site,
              obs_random_effects=obs, prior_site_re_sd=0.8, prior_obs_re_sd=1.2)
    return (g, oracle.OracleData(g["site_covs"], g["obs_covs"], g["obs"], **kw), OccuDataset(g["site_covs"], g["obs_covs"], g["obs"], **kw))


@pytest.mark.parametrize("name,site,obs,mode", CASES)
def test_cop_re_logp_grad_parity(name, site, obs, mode):
    """float32 kernel vs float64 oracle over every coordinate (the count model's tolerances: 2e-6 / 2e-5)."""
    _, od, ds = _pair(name, site, obs, mode)
    assert ds.D == od.D
    th = np.random.default_rng(4).uniform(-0.6, 0.6, size=(3, od.D)).astype(np.float32).astype(np.float64)
    Uo, Go = od.potential_grad(th)
    Ug, Gg = ds.logp_grad(th)
    assert np.all(np.isfinite(Ug)) and np.all(np.isfinite(Gg))
    assert np.max(np.abs(Ug - Uo) / np.abs(Uo)) <= 2e-6, (Ug, Uo)
    assert np.max(np.abs(Gg - Go)) <= 2e-5 * np.max(np.abs(Go)), np.max(np.abs(Gg - Go)) / np.max(np.abs(Go))


@pytest.mark.parametrize("k", [1, 3, 16])
@pytest.mark.parametrize("name,site,obs,mode", CASES[:3])
def test_cop_re_first_transitions_match_oracle(name, site, obs, mode, k):
    g, od, ds = _pair(name, site, obs, mode)
    init = np.random.default_rng(2).uniform(-0.5, 0.5, size=(2, od.D))   # (uniform(-2, 2) starts put rates at e^+-2 times the exposure: both sides start in the bulk)
    o = oracle.nuts_run(od, 0, 4, num_chains=2, seed=3, init=init)
    r = ds.nuts(num_warmup=0, num_samples=4, num_chains=2, seed=3, wgs_per_chain=k, init_theta=init)
    assert np.array_equal(o["num_steps"][:, :2], r.num_steps[:, :2]), (o["num_steps"], r.num_steps)
    assert np.allclose(o["draws"][:, 0], r.draws[:, 0], atol=5e-3)


def test_cop_re_posterior_matches_oracle():
    _, od, ds = _pair("cop_small_2x2", True, False, "constant")
    G = od.Ks + od.Ko + 3   # (the coefficients and phi = log rate_fp)
    o = oracle.nuts_run(od, 500, 1000, num_chains=4, seed=0)
    r = ds.nuts(num_warmup=500, num_samples=1000, num_chains=4, seed=50)
    posterior_parity(r.draws[:, :, :G], o["draws"][:, :, :G])
    sg, so = r.draws[:, :, G:G + 1].astype(np.float64), o["draws"][:, :, G:G + 1]
    mcse = np.sqrt(sg.var() / oracle.effective_sample_size(sg)[0] + so.var() / oracle.effective_sample_size(so)[0])
    assert abs(sg.mean() - so.mean()) <= 4 * mcse, (sg.mean(), so.mean(), mcse)


def _data():
    """As the reference's tests call it; the data dict carries false_positives_constant=True (occu_cop.py:386), so the three fits below
    sample the effects TOGETHER with a false-positive rate."""
    with contextlib.redirect_stdout(io.StringIO()):
        data, truth = simulate_cop(simulate_missing=True)
    assert data["false_positives_constant"] is True
    return data, truth


def test_reference_cop_site_random_effects():
    """occu_cop.py:473-493; predict() draws z and the counts with the effects in both predictors."""
    data, truth = _data()
    res = fit(occu_cop, **data, site_random_effects=True, num_chains=1, num_samples=500, timeout=600)
    s = res.samples
    assert "site_re_sd" in s and "site_re_occ" in s and "site_re_det" in s and "rate_fp_constant" in s
    assert s["site_re_sd"].mean() > 0 and 0.0 < s["rate_fp_constant"].mean() < 0.5
    assert np.allclose(s["psi"].mean(), truth["z"].mean(), atol=0.15)
    n_sites = data["obs"].shape[1]
    assert s["site_re_occ"].shape == (500, n_sites, 1) and s["rate_detection"].shape[0] == 500
    pred = predict(occu_cop, res.mcmc, **data, site_random_effects=True, num_samples=500)   # (data carries false_positives_constant)
    assert pred["z"].shape == (500, 1, n_sites, 1) and pred["y"].shape[0] == 500
    assert np.allclose(pred["psi"], s["psi"], rtol=1e-5)
    assert abs(pred["z"].mean() - s["psi"].mean()) < 0.03


def test_reference_cop_obs_random_effects():
    """occu_cop.py:496-515."""
    data, truth = _data()
    res = fit(occu_cop, **data, obs_random_effects=True, num_chains=1, num_samples=500, timeout=600)
    s = res.samples
    assert "obs_re_sd" in s and "obs_re" in s
    assert s["obs_re_sd"].mean() > 0
    assert np.allclose(s["psi"].mean(), truth["z"].mean(), atol=0.15)


def test_reference_cop_combined_random_effects():
    """occu_cop.py:518-545."""
    data, truth = _data()
    res = fit(occu_cop, **data, site_random_effects=True, obs_random_effects=True, num_chains=1, num_samples=500, timeout=600)
    s = res.samples
    for k in ("site_re_sd", "site_re_occ", "site_re_det", "obs_re_sd", "obs_re"):
        assert k in s
    assert np.allclose(s["psi"].mean(), truth["z"].mean(), atol=0.15)


def test_cop_re_without_a_false_positive_rate():
    """The effects alone: the same data less its false positives (the simulator's are the counts at unoccupied sites, occu_cop.py:343-357)."""
    data, truth = _data()
    data.pop("false_positives_constant")
    obs = data["obs"].copy()
    unocc = np.broadcast_to((truth["z"].transpose(0, 2, 1) == 0)[..., None], obs.shape)
    obs[unocc & np.isfinite(obs)] = 0.0
    res = fit(occu_cop, **{**data, "obs": obs}, site_random_effects=True, num_chains=1, num_samples=300, num_warmup=300, timeout=600)
    assert "rate_fp_constant" not in res.samples and np.allclose(res.samples["psi"].mean(), truth["z"].mean(), atol=0.15)
