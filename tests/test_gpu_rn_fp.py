"""Royle-Nichols model with a false-positive rate (biolith/models/occu_rn.py:133-138, 214-221): y ~ Bernoulli(1 - (1 - p)(1 - f)),
f ~ Beta(a, b); with and without the random effects.  theta = [beta, alpha, phi = logit f, (log sds), (effects)].  The kernels
(re_kernel.hpp, kind 5) through the C-ABI (bl_dataset_create_rn_fp) against the float64 oracle: potential + gradient over every
coordinate, the first trees on shared streams, the posterior, fit() and predict()."""
import contextlib
import io

import numpy as np
import pytest

import oracle
from biolith_amd.engine import OccuDataset
from biolith_amd.models import occu_rn, simulate_rn
from biolith_amd.utils import fit, predict
from conftest import load_golden, posterior_parity

pytestmark = pytest.mark.gpu

CASES = [("rn_small_2x2", 15, False, False), ("rn_small_2x2", 40, True, False), ("rn_missing", 25, True, True), ("rn_default", 100, False, False)]


def _pair(name, K, site, obs, **kw):
    g = load_golden(name)
    kw = dict(model="occu_rn", max_abundance=K, site_random_effects=site, obs_random_effects=obs, prior_site_re_sd=0.8, prior_obs_re_sd=1.2,
              re_fp_mode="constant", prior_fp=(2.0, 6.0), **kw)
    return (g, oracle.OracleData(g["site_covs"], g["obs_covs"], g["obs"], **kw), OccuDataset(g["site_covs"], g["obs_covs"], g["obs"], **kw))


@pytest.mark.parametrize("name,K,site,obs", CASES)
def test_rn_fp_logp_grad_parity(name, K, site, obs):
    _, od, ds = _pair(name, K, site, obs)
    assert ds.D == od.D
    th = np.random.default_rng(4).uniform(-0.8, 0.8, size=(3, od.D)).astype(np.float32).astype(np.float64)
    th[2, od.Ks + od.Ko + 2] = -4.0   # a small rate: the detection branch near the plain model's clamp
    Uo, Go = od.potential_grad(th)
    Ug, Gg = ds.logp_grad(th)
    assert np.all(np.isfinite(Ug)) and np.all(np.isfinite(Gg))
    assert np.max(np.abs(Ug - Uo) / np.abs(Uo)) <= 1e-5, (Ug, Uo)
    assert np.max(np.abs(Gg - Go)) <= 1e-4 * np.max(np.abs(Go)), np.max(np.abs(Gg - Go)) / np.max(np.abs(Go))
    fpi = od.Ks + od.Ko + 2      # the rate's own coordinate
    assert np.max(np.abs(Gg[:, fpi] - Go[:, fpi])) <= 1e-4 * np.max(np.abs(Go))


@pytest.mark.parametrize("k", [1, 3])
@pytest.mark.parametrize("name,K,site,obs", CASES[:3])
def test_rn_fp_first_transitions_match_oracle(name, K, site, obs, k):
    _, od, ds = _pair(name, K, site, obs)
    o = oracle.nuts_run(od, 0, 4, num_chains=2, seed=3)
    r = ds.nuts(num_warmup=0, num_samples=4, num_chains=2, seed=3, wgs_per_chain=k)
    assert np.array_equal(o["num_steps"][:, :2], r.num_steps[:, :2]), (o["num_steps"], r.num_steps)
    assert np.allclose(o["draws"][:, 0], r.draws[:, 0], atol=5e-3)


def test_rn_fp_posterior_matches_oracle():
    _, od, ds = _pair("rn_small_2x2", 15, False, False)
    o = oracle.nuts_run(od, 500, 1000, num_chains=4, seed=0)
    r = ds.nuts(num_warmup=500, num_samples=1000, num_chains=4, seed=50)
    posterior_parity(r.draws, o["draws"])


def test_fit_occu_rn_with_false_positives():
    with contextlib.redirect_stdout(io.StringIO()):
        data, truth = simulate_rn(n_sites=300, deployment_days_per_site=140, prob_fp=0.05, random_seed=2)
    res = fit(occu_rn, **data, false_positives_constant=True, num_chains=2, num_samples=400, num_warmup=400, timeout=600)
    s = res.samples
    assert s["prob_fp_constant"].shape == (800,) and 0.0 < float(s["prob_fp_constant"].mean()) < 0.2
    assert s["abundance"].shape == (800, 1, 300, 1)
    assert np.allclose(s["abundance"].mean(), truth["abundance"].mean(), rtol=0.25)
    pred = predict(occu_rn, res.mcmc, **data, false_positives_constant=True, num_samples=800)
    assert pred["N_i"].shape == (800, 1, 300, 1) and pred["y"].shape[0] == 800
    assert abs(pred["y"].mean() - np.nanmean(data["obs"])) < 0.05
    # ... and together with site random effects (every option of the reference's model at once)
    res2 = fit(occu_rn, **data, false_positives_constant=True, site_random_effects=True, num_chains=1, num_samples=20, num_warmup=20, timeout=600)
    for k in ("prob_fp_constant", "site_re_sd", "site_re_abu", "site_re_det"):
        assert k in res2.samples


def test_rn_fp_rejects_several_species():
    with contextlib.redirect_stdout(io.StringIO()):
        data, _ = simulate_rn(n_species=2, n_sites=30, random_seed=1)
    with pytest.raises(NotImplementedError):
        fit(occu_rn, **data, false_positives_constant=True, num_chains=1, num_samples=5, num_warmup=5)
