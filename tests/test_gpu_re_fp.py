"""Random effects TOGETHER with a false-positive rate (biolith/models/occu.py:146-157 with :170-173, 191-196): the reference
allows both flags at once.  theta = [beta, alpha, phi = logit(rate), (log sds), (effects)].  The kernels (re_kernel.hpp, kind 2)
through the C-ABI against the CPU oracle: potential + gradient over every coordinate, the first trees on shared streams (one and
several workgroups per chain), the posterior, and fit()'s sample sites."""
import contextlib
import io

import numpy as np
import pytest

import oracle
from biolith_amd.engine import OccuDataset
from biolith_amd.models import occu, simulate
from biolith_amd.utils import fit, predict
from conftest import load_golden, posterior_parity

pytestmark = pytest.mark.gpu

CASES = [("fp_constant", "constant", True, False), ("fp_constant", "constant", False, True), ("fp_unoccupied", "unoccupied", True, True),
         ("small_3x3", "unoccupied", True, False), ("missing", "constant", True, True)]


def _pair(name, mode, site, obs, **kw):
    g = load_golden(name)
    kw = dict(site_random_effects=site, obs_random_effects=obs, prior_site_re_sd=0.8, prior_obs_re_sd=1.2, re_fp_mode=mode, prior_fp=(2.0, 6.0), **kw)
    return (g, oracle.OracleData(g["site_covs"], g["obs_covs"], g["obs"], model="occu_re", **kw),
            OccuDataset(g["site_covs"], g["obs_covs"], g["obs"], model="occu_re", **kw))


@pytest.mark.parametrize("name,mode,site,obs", CASES)
def test_re_fp_logp_grad_parity(name, mode, site, obs):
    """float32 kernel vs float64 oracle over every coordinate (the random-effects model's tolerances: 2e-6 / 2e-5)."""
    _, od, ds = _pair(name, mode, site, obs)
    assert ds.D == od.D
    th = np.random.default_rng(4).uniform(-1.2, 1.2, size=(3, od.D)).astype(np.float32).astype(np.float64)
    Uo, Go = od.potential_grad(th)
    Ug, Gg = ds.logp_grad(th)
    assert np.all(np.isfinite(Ug)) and np.all(np.isfinite(Gg))
    assert np.max(np.abs(Ug - Uo) / np.abs(Uo)) <= 2e-6, (Ug, Uo)
    assert np.max(np.abs(Gg - Go)) <= 2e-5 * np.max(np.abs(Go)), np.max(np.abs(Gg - Go)) / np.max(np.abs(Go))
    fpi = od.Ks + od.Ko + 2      # the rate's own coordinate
    assert np.max(np.abs(Gg[:, fpi] - Go[:, fpi])) <= 2e-5 * np.max(np.abs(Go))


@pytest.mark.parametrize("k", [1, 3, 16])
@pytest.mark.parametrize("name,mode,site,obs", CASES[:3])
def test_re_fp_first_transitions_match_oracle(name, mode, site, obs, k):
    _, od, ds = _pair(name, mode, site, obs)
    o = oracle.nuts_run(od, 0, 4, num_chains=2, seed=3)
    r = ds.nuts(num_warmup=0, num_samples=4, num_chains=2, seed=3, wgs_per_chain=k)
    assert np.array_equal(o["num_steps"][:, :2], r.num_steps[:, :2]), (o["num_steps"], r.num_steps)
    assert np.allclose(o["draws"][:, 0], r.draws[:, 0], atol=5e-3)


def test_re_fp_adaptation_and_next_tree_match_oracle():
    g, od, ds = _pair("fp_constant", "constant", True, False)
    init = np.zeros((2, od.D))
    init[:, :od.Ks + od.Ko + 2] = np.concatenate([g["beta"][0], g["alpha"][0]])
    init[:, od.Ks + od.Ko + 2] = np.log(0.1 / 0.9)
    o = oracle.nuts_run(od, 8, 3, num_chains=2, seed=5, init=init)
    r = ds.nuts(num_warmup=8, num_samples=3, num_chains=2, seed=5, init_theta=init)
    assert np.allclose(o["step_size"], r.step_size, rtol=2e-3)
    assert np.array_equal(o["num_steps"][:, :1], r.num_steps[:, :1])


def test_re_fp_posterior_matches_oracle():
    """The false-positive likelihood is bimodal (test_gpu_fp.py): both samplers start at the generating parameters."""
    g, od, ds = _pair("fp_constant", "constant", True, False)
    G = od.Ks + od.Ko + 2
    init = np.zeros((4, od.D))
    init[:, :G] = np.concatenate([g["beta"][0], g["alpha"][0]])
    init[:, G] = np.log(0.1 / 0.9)
    init[:, G + 1] = np.log(0.3)
    o = oracle.nuts_run(od, 500, 1000, num_chains=4, seed=0, init=init)
    r = ds.nuts(num_warmup=500, num_samples=1000, num_chains=4, seed=50, init_theta=init)
    # the fixed effects and the rate at SURVEY 8c's tolerances; log site_re_sd sits at the neck of the centred parameterisation's
    # funnel (the reference's own parameterisation; test_gpu_re.py) and is only required to agree with the oracle's draws in mean
    posterior_parity(r.draws[:, :, :G + 1], o["draws"][:, :, :G + 1])
    sg, so = r.draws[:, :, G + 1:G + 2].astype(np.float64), o["draws"][:, :, G + 1:G + 2]
    mcse = np.sqrt(sg.var() / oracle.effective_sample_size(sg)[0] + so.var() / oracle.effective_sample_size(so)[0])
    assert abs(sg.mean() - so.mean()) <= 4 * mcse, (sg.mean(), so.mean(), mcse)


def test_fit_occu_with_random_effects_and_false_positives():
    with contextlib.redirect_stdout(io.StringIO()):
        data, truth = simulate(n_sites=300, deployment_days_per_site=84, prob_fp_constant=0.1, site_random_effects=True, random_seed=3)
    res = fit(occu, **data, false_positives_constant=True, site_random_effects=True, num_chains=4, num_samples=400, num_warmup=400)
    s = res.samples
    assert s["prob_fp_constant"].shape == (1600,) and s["site_re_sd"].shape == (1600,)
    assert s["site_re_occ"].shape == (1600, 300, 1) and s["psi"].shape == (1600, 1, 300, 1)
    assert res.mcmc.result.draws.shape[2] == 4 + 1 + 1 + 600
    assert 0.0 < float(s["prob_fp_constant"].mean()) < 0.35
    assert abs(float(s["psi"].mean()) - truth["z"].mean()) < 0.15
    assert s["prob_detection_fp"].shape[:3] == (1600, 2, 12)
    # predict(): z and y drawn with the effects in both predictors and the rate on top (occu.py:229-241)
    pred = predict(occu, res.mcmc, **data, false_positives_constant=True, site_random_effects=True, num_samples=1600)
    assert pred["z"].shape == (1600, 1, 300, 1) and pred["y"].shape == (1600, 12, 1, 300, 1)
    assert np.allclose(pred["psi"], s["psi"], rtol=1e-5)
    assert abs(pred["z"].mean() - s["psi"].mean()) < 0.02
    # P(y = 1) = 1 - (1 - z p)(1 - f): its mean over draws, sites and visits against the data's detection frequency
    assert abs(pred["y"].mean() - np.nanmean(data["obs"])) < 0.03
    zp = pred["prob_detection"] * pred["z"][:, None].astype(np.float32)
    f = s["prob_fp_constant"].reshape(-1, 1, 1, 1, 1)
    assert abs(pred["y"].mean() - float((1.0 - (1.0 - zp) * (1.0 - f)).mean())) < 0.01


def test_re_fp_rejects_several_species():
    with contextlib.redirect_stdout(io.StringIO()):
        data, _ = simulate(n_species=2, n_sites=30, random_seed=1)
    with pytest.raises(NotImplementedError):
        fit(occu, **data, false_positives_constant=True, site_random_effects=True, num_chains=1, num_samples=5, num_warmup=5)
