"""Dynamic occupancy kernel (BASELINE.json configs[4] as worded: colonisation / extinction, forward algorithm) through the C-ABI
against the builder's own float64 oracle.  NO REFERENCE COUNTERPART: timmh/biolith has no such model, so
every comparison here is HIP vs oracle/occu_oracle.c (pinned by brute force over the latent paths, tests/test_dyn_cpu.py)."""
import contextlib
import io

import numpy as np
import pytest

import oracle
from biolith_amd.engine import OccuDataset
from biolith_amd.models import occu_dyn, simulate_dyn
from biolith_amd.utils import fit
from conftest import PARITY_S, PARITY_W, posterior_parity

pytestmark = pytest.mark.gpu


def _sim(**kw):
    with contextlib.redirect_stdout(io.StringIO()):
        return simulate_dyn(**kw)


def _pair(data, **kw):
    return (oracle.OracleData(data["site_covs"], data["obs_covs"], data["obs"], model="occu_dyn", **kw),
            OccuDataset(data["site_covs"], data["obs_covs"], data["obs"], model="occu_dyn", **kw))


@pytest.mark.parametrize("cfg", [dict(n_sites=150, n_periods=5), dict(n_sites=301, n_periods=3, n_site_covs=2, n_obs_covs=3, simulate_missing=True),
                                 dict(n_sites=64, n_periods=1), dict(n_sites=97, n_periods=12, n_site_covs=3, n_obs_covs=1, deployment_days_per_site=14)])
def test_dyn_logp_grad_parity(cfg):
    """float32 kernel vs float64 oracle: |dU| / |U| <= 1e-6, max |dgrad| <= 1e-5 max |grad| (the occu kernel's tolerances)."""
    data, _ = _sim(random_seed=3, **cfg)
    od, ds = _pair(data, prior_beta=(0.1, 1.5), prior_alpha=(-0.2, 0.7))
    assert ds.D == od.D
    th = np.random.default_rng(4).uniform(-1.5, 1.5, size=(5, od.D)).astype(np.float32).astype(np.float64)
    Uo, Go = od.potential_grad(th)
    Ug, Gg = ds.logp_grad(th)
    assert np.all(np.isfinite(Ug)) and np.all(np.isfinite(Gg))
    assert np.max(np.abs(Ug - Uo) / np.abs(Uo)) <= 1e-6, (Ug, Uo)
    assert np.max(np.abs(Gg - Go)) <= 1e-5 * np.max(np.abs(Go)), np.max(np.abs(Gg - Go)) / np.max(np.abs(Go))


@pytest.mark.parametrize("T", [2, 4, 8])
def test_dyn_two_scans_form(T, monkeypatch):
    """One period per lane (lanes per pair == periods: dyn_device.hpp bl_eval_sites_dyn_scan, round 5 -- the forward-backward algorithm
    as a prefix and a suffix scan of 2 x 2 matrices over the group's lanes): K1 at the occu kernel's tolerances, also with missing
    visits, an odd site count, at coefficients far out (|logit| up to ~12: the products' power-of-two rescaling), and the oracle's
    first trees."""
    data, _ = _sim(random_seed=5 + T, n_sites=203, n_periods=T, n_site_covs=2, n_obs_covs=2, simulate_missing=True)
    monkeypatch.setenv("BIOLITH_HIP_DYN_G", str(T))
    od, ds = _pair(data)
    rng = np.random.default_rng(T)
    th = rng.uniform(-1.5, 1.5, size=(6, od.D))
    th[4] *= 3.0                      # far out: transition probabilities down to ~1e-5
    th[5, [0, 3, 6]] = [6.0, -7.0, -7.0]   # psi ~ 1, gamma ~ eps ~ 1e-3
    th = th.astype(np.float32).astype(np.float64)
    Uo, Go = od.potential_grad(th)
    Ug, Gg = ds.logp_grad(th)
    assert np.all(np.isfinite(Ug)) and np.all(np.isfinite(Gg))
    assert np.max(np.abs(Ug - Uo) / np.abs(Uo)) <= 1e-6, (Ug, Uo)
    assert np.max(np.abs(Gg - Go) / np.max(np.abs(Go), axis=1, keepdims=True)) <= 1e-5, np.abs(Gg - Go).max(1)
    o = oracle.nuts_run(od, 0, 4, num_chains=2, seed=T)
    r = ds.nuts(num_warmup=0, num_samples=4, num_chains=2, seed=T)
    assert r.kernel_name.rstrip().endswith((", true, 1, false>", ", true, 2, false>")), r.kernel_name   # a two-scans instantiation ran (2: 8 x 4 as compile-time facts)
    assert np.array_equal(o["num_steps"][:, :3], r.num_steps[:, :3]), (o["num_steps"], r.num_steps)
    assert np.allclose(o["draws"][:, 0], r.draws[:, 0], atol=2e-3)
    ds.close()


def test_dyn_first_transitions_match_oracle():
    data, _ = _sim(n_sites=200, n_periods=6, random_seed=1)
    od, ds = _pair(data)
    o = oracle.nuts_run(od, 12, 8, num_chains=3, seed=5)
    r = ds.nuts(num_warmup=12, num_samples=8, num_chains=3, seed=5)
    assert np.array_equal(o["num_steps"][:, :4], r.num_steps[:, :4]), (o["num_steps"], r.num_steps)
    assert (o["num_steps"] == r.num_steps).mean() >= 0.8
    assert np.allclose(o["draws"][:, 0], r.draws[:, 0], atol=2e-2)
    assert np.allclose(o["step_size"], r.step_size, rtol=0.05)


def test_dyn_posterior_matches_oracle():
    data, _ = _sim(n_sites=300, n_periods=6, random_seed=1)
    od, ds = _pair(data)
    o = oracle.nuts_run(od, PARITY_W, PARITY_S, num_chains=4, seed=0)
    r = ds.nuts(num_warmup=PARITY_W, num_samples=PARITY_S, num_chains=4, seed=100)
    assert r.diverging.sum() == 0
    posterior_parity(r.draws, o["draws"])


def test_dyn_config5_full_size():
    """BASELINE.json configs[4]: 2 000 sites x 8 seasons x 4 visits, 3 + 3 covariates, 4 chains -- K1 and the sampler's first
    trees against the oracle at full size, then recovery of the generating rates through fit()."""
    data, truth = _sim(n_sites=2000, n_periods=8, n_site_covs=3, n_obs_covs=3, deployment_days_per_site=28, session_duration=7)
    od, ds = _pair(data)
    assert (ds.N, ds.T, ds.J, ds.D) == (2000, 8, 4, 16)
    th = np.random.default_rng(1).uniform(-2, 2, size=(4, od.D)).astype(np.float32).astype(np.float64)
    Uo, Go = od.potential_grad(th)
    Ug, Gg = ds.logp_grad(th)
    assert np.max(np.abs(Ug - Uo) / np.abs(Uo)) <= 1e-6 and np.max(np.abs(Gg - Go)) <= 1e-5 * np.max(np.abs(Go))
    o = oracle.nuts_run(od, 10, 4, num_chains=2, seed=3)
    r = ds.nuts(num_warmup=10, num_samples=4, num_chains=2, seed=3)
    assert r.lds_staged
    assert np.array_equal(o["num_steps"][:, :3], r.num_steps[:, :3]), (o["num_steps"], r.num_steps)
    assert np.allclose(o["step_size"], r.step_size, rtol=0.02)
    res = fit(occu_dyn, **data, num_chains=4, num_warmup=500, num_samples=500)
    assert res.mcmc.result.diverging.sum() == 0
    for name in ("psi", "gamma", "epsilon"):
        assert res.samples[name].shape == (2000, 2000, 1)
        assert abs(float(res.samples[name].mean()) - truth[name].mean()) < 0.05, name
    want = np.concatenate([truth["beta"][0], truth["beta_col"][0], truth["beta_ext"][0], truth["alpha"][0]])
    est = res.mcmc.result.draws.reshape(-1, 16).mean(0)
    assert np.abs(est - want).max() < 0.5          # the reference's coefficient tolerance (occu.py:440-456)
    print("cfg5 (dynamic) kernel ms", res.mcmc.result.kernel_ms, "us/leapfrog/chain",
          res.mcmc.result.kernel_ms * 1e3 / (res.mcmc.result.n_leapfrog.sum() / 4))


def test_fit_occu_dyn_names_and_shapes():
    data, truth = _sim(n_sites=120, n_periods=4, n_site_covs=2, n_obs_covs=1, simulate_missing=True, random_seed=4)
    res = fit(occu_dyn, **data, num_chains=2, num_warmup=200, num_samples=150)
    s = res.samples
    for k in ("cov_state_0", "cov_state_2", "cov_col_0", "cov_col_2", "cov_ext_1", "cov_det_0", "cov_det_1"):
        assert s[k].shape == (300, 1), k
    assert "cov_det_2" not in s and s["gamma"].shape == (300, 120, 1)
    assert res.mcmc.result.draws.shape == (2, 150, 3 * 3 + 2)


def test_dyn_rejects_what_is_not_built():
    data, _ = _sim(n_sites=40, n_periods=3, n_site_covs=9, random_seed=2)
    with pytest.raises(NotImplementedError):
        fit(occu_dyn, **data)


def test_predict_for_the_dynamic_model():
    """Posterior predictive sites of the builder-defined model (no reference counterpart): exact identities and 5-sigma frequencies."""
    from biolith_amd.utils import predict

    data, truth = _sim(n_sites=150, n_periods=5, n_site_covs=2, n_obs_covs=1, random_seed=6)
    res = fit(occu_dyn, **data, num_chains=2, num_warmup=200, num_samples=400)
    pp = predict(occu_dyn, res.mcmc, **data, random_seed=3)
    n, N, T, J = 800, 150, 5, data["obs"].shape[3]
    assert pp["psi"].shape == pp["gamma"].shape == pp["epsilon"].shape == (n, N, 1)
    assert pp["z"].shape == (n, T, N, 1) and pp["prob_detection"].shape == pp["y"].shape == (n, J, T, N, 1)
    assert np.array_equal(pp["psi"], res.samples["psi"]) and np.array_equal(pp["gamma"], res.samples["gamma"])   # the fit's own deterministic sites
    z, y, p = pp["z"][..., 0], pp["y"][..., 0], pp["prob_detection"][..., 0]
    assert set(np.unique(z)) <= {0, 1} and np.all(y[(z[:, None] == 0) & np.ones_like(y, bool)] == 0)   # no detection at an unoccupied site
    # z_1 ~ Bernoulli(psi); z_t+1 | z_t ~ Bernoulli(gamma) / Bernoulli(1 - epsilon); y | z = 1 ~ Bernoulli(p): frequencies within 5 sigma
    def close(freq, prob, count):
        return abs(freq - prob) < 5 * np.sqrt(prob * (1 - prob) / count) + 1e-3
    psi, gam, eps = pp["psi"][..., 0], pp["gamma"][..., 0], pp["epsilon"][..., 0]
    assert close(z[:, 0].mean(), psi.mean(), n * N)
    was0, was1 = z[:, :-1] == 0, z[:, :-1] == 1
    g_b = np.broadcast_to(gam[:, None], z[:, 1:].shape)
    e_b = np.broadcast_to(eps[:, None], z[:, 1:].shape)
    assert close(z[:, 1:][was0].mean(), g_b[was0].mean(), was0.sum())
    assert close(z[:, 1:][was1].mean(), 1.0 - e_b[was1].mean(), was1.sum())
    occ = np.broadcast_to(z[:, None] == 1, y.shape)
    assert close(y[occ].mean(), p[occ].mean(), occ.sum())
    again = predict(occu_dyn, res.mcmc, **data, random_seed=3)
    assert np.array_equal(again["z"], pp["z"]) and np.array_equal(again["y"], pp["y"])   # keyed by the seed
    assert not np.array_equal(predict(occu_dyn, res.mcmc, **data, random_seed=4)["z"], pp["z"])
