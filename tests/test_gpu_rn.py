"""Royle-Nichols kernel (config 4 family) through the C-ABI against the CPU oracle, plus the reference's
own fit(occu_rn) assertions (biolith/models/occu_rn.py:361-407)."""
import numpy as np
import pytest

import oracle
from biolith_amd.engine import OccuDataset
from biolith_amd.evaluation import split_gelman_rubin
from biolith_amd.models import occu_rn, simulate_rn
from biolith_amd.utils import fit
from conftest import PARITY_S, PARITY_W, load_golden, posterior_parity

pytestmark = pytest.mark.gpu


def _pair(name, **kw):
    g = load_golden(name)
    return (g, oracle.OracleData(g["site_covs"], g["obs_covs"], g["obs"], model="occu_rn", **kw),
            OccuDataset(g["site_covs"], g["obs_covs"], g["obs"], model="occu_rn", **kw))


@pytest.mark.parametrize("name", ["rn_small_2x2", "rn_missing", "rn_default"])
def test_rn_logp_grad_parity(name):
    """float32 kernel vs float64 oracle: |dU|/|U| <= 1e-5, max|dgrad| <= 1e-4 max|grad| (the 101-term
    sums over N accumulate more float32 rounding than the occu kernel's)."""
    _, od, ds = _pair(name)
    th = np.random.default_rng(4).uniform(-1.2, 1.2, size=(4, od.D)).astype(np.float32).astype(np.float64)
    Uo, Go = od.potential_grad(th)
    Ug, Gg = ds.logp_grad(th)
    assert np.all(np.isfinite(Ug)) and np.all(np.isfinite(Gg))
    assert np.max(np.abs(Ug - Uo) / np.abs(Uo)) <= 1e-5, (Ug, Uo)
    assert np.max(np.abs(Gg - Go)) <= 1e-4 * np.max(np.abs(Go)), np.max(np.abs(Gg - Go))


def test_rn_small_cutoff_and_priors():
    g = load_golden("rn_small_2x2")
    th = np.random.default_rng(1).uniform(-1, 1, size=(2, 6)).astype(np.float32).astype(np.float64)
    # K <= 103 runs the 104-entry table instantiation, larger K the 128-entry one (occu_device.hpp); 5, 6, 7, 8: the model's
    # bound n <= K inside a block of four unrolled terms
    for K, pri in ((5, ((0.0, 1.0), (0.0, 1.0))), (6, ((0.0, 1.0), (0.0, 1.0))), (7, ((0.0, 1.0), (0.0, 1.0))), (8, ((0.0, 1.0), (0.0, 1.0))),
                   (40, ((0.3, 2.0), (-0.2, 0.5))), (103, ((0.0, 1.0), (0.0, 1.0))),
                   (104, ((0.0, 1.0), (0.0, 1.0))), (127, ((0.1, 1.5), (0.0, 1.0)))):
        od = oracle.OracleData(g["site_covs"], g["obs_covs"], g["obs"], *pri, model="occu_rn", max_abundance=K)
        ds = OccuDataset(g["site_covs"], g["obs_covs"], g["obs"], *pri, model="occu_rn", max_abundance=K)
        Uo, Go = od.potential_grad(th)
        Ug, Gg = ds.logp_grad(th)
        assert np.max(np.abs(Ug - Uo) / np.abs(Uo)) <= 1e-5
        assert np.max(np.abs(Gg - Go)) <= 1e-4 * np.max(np.abs(Go))


def test_rn_first_transitions_match_oracle():
    _, od, ds = _pair("rn_small_2x2")
    o = oracle.nuts_run(od, 0, 5, num_chains=2, seed=3)
    r = ds.nuts(num_warmup=0, num_samples=5, num_chains=2, seed=3)
    assert np.array_equal(o["num_steps"][:, :3], r.num_steps[:, :3]), (o["num_steps"], r.num_steps)
    assert np.allclose(o["draws"][:, 0], r.draws[:, 0], atol=2e-3)


def test_rn_posterior_matches_oracle():
    _, od, ds = _pair("rn_small_2x2")
    o = oracle.nuts_run(od, PARITY_W, PARITY_S, num_chains=4, seed=0)
    r = ds.nuts(num_warmup=PARITY_W, num_samples=PARITY_S, num_chains=4, seed=50)
    posterior_parity(r.draws, o["draws"])


def test_occu_rn_like_reference():  # occu_rn.py:361-388 (simulate_rn default: 100 sites x 52 visits)
    data, true_params = simulate_rn(simulate_missing=True)
    results = fit(occu_rn, **data, timeout=600)
    assert np.allclose(results.samples["abundance"].mean(), true_params["abundance"].mean(), rtol=0.1)
    assert np.allclose([results.samples[k].mean() for k in [f"cov_state_{i}" for i in range(true_params["beta"].shape[1])]],
                       true_params["beta"].mean(axis=0), atol=0.5)
    assert np.allclose([results.samples[k].mean() for k in [f"cov_det_{i}" for i in range(true_params["alpha"].shape[1])]],
                       true_params["alpha"].mean(axis=0), atol=0.5)
    assert results.samples["abundance"].shape == (5000, 1, 100, 1)
    assert "psi" not in results.samples


def test_occu_rn_multi_season():  # occu_rn.py:391-407
    data, true_params = simulate_rn(simulate_missing=True, n_periods=3)
    results = fit(occu_rn, **data, num_chains=1, num_samples=300, num_warmup=300, timeout=600)
    assert np.allclose(results.samples["abundance"].mean(), true_params["abundance"].mean(), rtol=0.2)


def test_occu_rn_multi_species():  # occu_rn.py:412-420 (the reference's own test: a shape assert)
    data, _ = simulate_rn(simulate_missing=True, n_species=2, n_sites=30)
    results = fit(occu_rn, **data, num_chains=1, num_samples=200, timeout=600)
    assert results.samples["abundance"].shape[-1] == 2
    assert results.samples["abundance"].shape == (200, 1, 30, 2) and results.samples["cov_state_0"].shape == (200, 2)
    assert np.all(np.isfinite(results.samples["abundance"])) and not results.mcmc.get_extra_fields()["diverging"].all()


def test_rn_config4_runs_and_recovers_truth():
    """BASELINE config 4: 5 000 sites x 10 visits, 3+3 covariates, 4 chains."""
    data, truth = simulate_rn(n_sites=5000, n_site_covs=3, n_obs_covs=3, deployment_days_per_site=70, session_duration=7)
    res = fit(occu_rn, **data, num_chains=4, num_warmup=500, num_samples=500)
    assert np.allclose(res.samples["abundance"].mean(), truth["abundance"].mean(), rtol=0.1)  # occu_rn.py:368-372
    est = np.array([res.samples[f"cov_state_{i}"].mean() for i in range(4)] + [res.samples[f"cov_det_{i}"].mean() for i in range(4)])
    assert np.allclose(est, np.concatenate([truth["beta"][0], truth["alpha"][0]]), atol=0.15)
    assert split_gelman_rubin(res.mcmc.get_samples(group_by_chain=True)["beta"]).max() < 1.05
    print("cfg4 kernel ms", res.mcmc.result.kernel_ms, "leapfrogs", res.mcmc.result.n_leapfrog.sum(),
          "us/leapfrog/chain", res.mcmc.result.kernel_ms * 1e3 / (res.mcmc.result.n_leapfrog.sum() / 4))


def _cfg4():
    import contextlib
    import io

    with contextlib.redirect_stdout(io.StringIO()):
        return simulate_rn(n_sites=5000, n_site_covs=3, n_obs_covs=3, deployment_days_per_site=70, session_duration=7)


def test_rn_config4_full_size_against_oracle():
    """BASELINE config 4 at full size against the float64 oracle (one oracle evaluation costs about 50 ms here): K1 at the stated
    1e-5 / 1e-4 near the generating parameters and over init_to_uniform's box, then the sampler on shared streams -- the step
    sizes after ten adaptation steps from the generating parameters, the next trees and the first draw."""
    data, truth = _cfg4()
    od = oracle.OracleData(data["site_covs"], data["obs_covs"], data["obs"], model="occu_rn")
    ds = OccuDataset(data["site_covs"], data["obs_covs"], data["obs"], model="occu_rn")
    th0 = np.concatenate([truth["beta"][0], truth["alpha"][0]]).astype(np.float32).astype(np.float64)
    th = np.concatenate([th0[None] + np.random.default_rng(0).normal(0, 0.1, size=(3, 8)), np.random.default_rng(1).uniform(-2, 2, size=(3, 8))])
    th = th.astype(np.float32).astype(np.float64)
    Uo, Go = od.potential_grad(th)
    Ug, Gg = ds.logp_grad(th)
    assert np.max(np.abs(Ug - Uo) / np.abs(Uo)) <= 1e-5, (Ug, Uo)
    assert np.all(np.abs(Gg - Go).max(1) <= 1e-4 * np.abs(Go).max(1)), np.abs(Gg - Go).max(1) / np.abs(Go).max(1)
    init = np.tile(th0, (2, 1))
    o = oracle.nuts_run(od, 10, 3, num_chains=2, seed=3, init=init)
    r = ds.nuts(num_warmup=10, num_samples=3, num_chains=2, seed=3, init_theta=init)
    assert r.lds_staged and r.wgs_per_chain >= 16
    assert np.allclose(o["step_size"], r.step_size, rtol=5e-3), (o["step_size"], r.step_size)
    assert np.array_equal(o["num_steps"][:, :2], r.num_steps[:, :2]), (o["num_steps"], r.num_steps)
    assert np.allclose(o["draws"][:, 0], r.draws[:, 0], atol=5e-3)


def test_rn_config4_posterior_against_oracle_fixture():
    """4 chains x (500 + 500) at full size against the oracle's captured posterior (tests/golden/oracle_posterior_cfg4.json, made
    by make_oracle_posterior.py cfg4: a quarter of an hour of CPU, so a fixture)."""
    import json
    import os

    from conftest import GOLDEN

    data, truth = _cfg4()
    fx = json.load(open(os.path.join(GOLDEN, "oracle_posterior_cfg4.json")))
    ds = OccuDataset(data["site_covs"], data["obs_covs"], data["obs"], model="occu_rn")
    r = ds.nuts(num_warmup=fx["num_warmup"], num_samples=fx["num_samples"], num_chains=4, seed=0)
    assert r.diverging.sum() == 0
    flat = r.draws.reshape(-1, 8).astype(np.float64)
    from biolith_amd.evaluation import effective_sample_size

    mcse = np.sqrt(flat.var(0) / effective_sample_size(r.draws) + np.array(fx["sd"]) ** 2 / np.array(fx["ess"]))
    assert np.all(np.abs(flat.mean(0) - fx["mean"]) <= 4 * mcse), (flat.mean(0) - fx["mean"], mcse)
    assert np.all(np.abs(flat.std(0) / fx["sd"] - 1) < 0.1), flat.std(0) / fx["sd"]
    assert np.all(np.abs(flat.mean(0) - fx["map"]) < 3 * np.array(fx["laplace_sd"]))
    assert split_gelman_rubin(r.draws).max() < 1.01
    assert abs(r.num_steps.mean() / fx["mean_num_steps"] - 1) < 0.15
    assert abs(np.log(r.step_size.mean() / np.mean(fx["step_size"]))) < 0.2
    X = data["site_covs"].astype(np.float32).astype(np.float64)
    lam_mean = np.exp(flat[::10, :1] + flat[::10, 1:4] @ X.T).mean()
    assert abs(lam_mean - fx["psi_mean"]) < 4 * fx["psi_mean_sd"] / np.sqrt(100) + 2e-3    # (psi_mean: the mean abundance here)
    assert np.allclose(lam_mean, truth["abundance"].mean(), rtol=0.1)  # occu_rn.py:368-372


def test_rn_nondetection_clamp_regime():
    """numpyro floors a non-detection's n log(1-r) at log(eps_f32) = -15.94 (Bernoulli probabilities are clamped to
    [tiny, 1 - eps]); the kernel and the oracle both carry the floor.  Near the generating parameters of a config-4
    style dataset it is never reached with weight; at parameters that force N ~ 50 onto sites with non-detections it
    decides the potential (several per cent of it), and the two must still agree -- value and gradient."""
    from conftest import quiet_simulate  # noqa: F401  (same helper family)
    import contextlib
    import io

    with contextlib.redirect_stdout(io.StringIO()):
        data, truth = simulate_rn(n_sites=600, n_site_covs=3, n_obs_covs=3, deployment_days_per_site=70, session_duration=7, random_seed=1)
    od = oracle.OracleData(data["site_covs"], data["obs_covs"], data["obs"], model="occu_rn")
    ds = OccuDataset(data["site_covs"], data["obs_covs"], data["obs"], model="occu_rn")
    th0 = np.concatenate([truth["beta"][0], truth["alpha"][0]]).astype(np.float32).astype(np.float64)
    near = th0[None] + np.random.default_rng(0).normal(0, 0.1, size=(4, 8)).astype(np.float32)
    Uo, Go = od.potential_grad(near)
    Ug, Gg = ds.logp_grad(near)
    assert np.max(np.abs(Ug - Uo) / np.abs(Uo)) <= 1e-5
    assert np.max(np.abs(Gg - Go)) <= 1e-4 * np.max(np.abs(Go))
    far = np.array([[0.8, 1.0, 0.9, -0.8, 0.6, 0.2, 0.3, -0.1],        # abundances up to ~e^4 at the covariate tails
                    [1.5, 1.2, -1.0, 0.9, 1.8, 0.5, -0.4, 0.3],       # ... with high detection: every non-detection floored
                    [2.0, -2.0, 2.0, -2.0, 2.0, 2.0, -2.0, 2.0]])     # a corner of init_to_uniform's box
    Uo, Go = od.potential_grad(far)
    Ug, Gg = ds.logp_grad(far)
    assert np.all(np.isfinite(Ug))
    assert np.max(np.abs(Ug - Uo) / np.abs(Uo)) <= 2e-5, (Ug, Uo)
    assert np.max(np.abs(Gg - Go)) <= 1e-3 * np.max(np.abs(Go)), np.max(np.abs(Gg - Go)) / np.max(np.abs(Go))


@pytest.mark.parametrize("ks,ko", [(4, 4), (4, 2), (1, 4)])
def test_rn_covariate_capacities(ks, ko):
    """Every padded capacity pair has its own instantiation of the two occu_rn kernels (register allocation differs: the
    (4,4) one is the fullest): K1 parity, same first trees as the oracle, and a sane warmed-up sampler on each."""
    import contextlib
    import io

    with contextlib.redirect_stdout(io.StringIO()):
        d, t = simulate_rn(n_sites=300, n_site_covs=ks, n_obs_covs=ko, deployment_days_per_site=56, session_duration=7, random_seed=2)
    od = oracle.OracleData(d["site_covs"], d["obs_covs"], d["obs"], model="occu_rn")
    ds = OccuDataset(d["site_covs"], d["obs_covs"], d["obs"], model="occu_rn")
    th = np.random.default_rng(0).uniform(-0.6, 0.6, size=(3, od.D)).astype(np.float32).astype(np.float64)
    Uo, Go = od.potential_grad(th)
    Ug, Gg = ds.logp_grad(th)
    assert np.max(np.abs(Ug - Uo) / np.abs(Uo)) <= 1e-5 and np.max(np.abs(Gg - Go)) <= 1e-4 * np.max(np.abs(Go))
    o = oracle.nuts_run(od, 0, 4, num_chains=2, seed=3)
    r = ds.nuts(num_warmup=0, num_samples=4, num_chains=2, seed=3)
    assert np.array_equal(o["num_steps"][:, :3], r.num_steps[:, :3])
    r = ds.nuts(num_warmup=150, num_samples=150, num_chains=2, seed=1)
    assert r.num_steps.mean() < 40 and np.all(r.step_size > 0.05) and r.diverging.sum() == 0
    want = np.concatenate([t["beta"][0], t["alpha"][0]])
    assert np.abs(r.draws.reshape(-1, od.D).mean(0) - want).max() < 0.5   # the reference's coefficient tolerance


def _rn_data(rng, n_sites, n_visits, share_without_detection, n_periods=1, r_hi=False):
    """Royle-Nichols data by construction: a share of the sites has no detection at all (closed form in the kernel; the last compute
    wave's share of a workgroup when the split is on), the others at least one.  r_hi: detection probabilities near one at a few
    non-detections, so that numpyro's floor of n log q matters (occu_rn.py:216-219 through clamp_probs)."""
    X = rng.normal(size=(n_sites, 2)).astype(np.float32)
    W = rng.normal(size=(n_sites, n_periods, n_visits, 2)).astype(np.float32)
    Y = (rng.uniform(size=(1, n_sites, n_periods, n_visits)) < 0.35) * 1.0
    none = rng.uniform(size=n_sites) < share_without_detection
    Y[0, none] = 0.0
    some = ~none & (Y[0].reshape(n_sites, -1).sum(1) == 0)
    Y[0, some, 0, 0] = 1.0
    Y[rng.uniform(size=Y.shape) < 0.05] = np.nan
    if r_hi:
        W[:: 7, :, 1, :] = 6.0      # |nu| large at one visit of every seventh site
    return X, W, Y.astype(np.float32)


@pytest.mark.parametrize("n_sites,share,n_periods,r_hi,k", [
    (5000, 0.34, 1, False, 0),    # config 4's proportions: the split is on (32 workgroups x (4 x 26 + 53))
    (5000, 0.34, 1, True, 0),     # ... with floored non-detections within reach
    (5000, 0.0, 1, False, 0),     # every site with a detection: no split (equal shares for five waves)
    (5000, 1.0, 1, False, 0),     # no detection anywhere: no split, every site in closed form, no item round at all
    (5000, 0.05, 1, False, 0),    # few sites without one: their wave is nearly idle -- the detected ones do not fit four waves at two lanes: no split
    (5000, 0.9, 1, False, 0),     # mostly without
    (700, 0.4, 3, False, 0),      # several periods: a site counts as detected if any period has one; closed form per (site, period)
    (40000, 0.34, 1, False, 0),   # more than 508 sites per workgroup: no order table, plain shares
    (313, 0.3, 1, True, 2),       # two workgroups, odd counts
])
def test_rn_wave_shares_and_closed_form(n_sites, share, n_periods, r_hi, k):
    """Round 6 (rn_device.hpp): a workgroup stages its sites with a detection first and gives the others -- closed form, no sum over n --
    to its last compute wave; the first visit to floor is stated without a branch.  K1 against the float64 oracle on every branch of
    that logic, and a short sampler run that must build the oracle's first trees."""
    rng = np.random.default_rng(n_sites + int(100 * share) + n_periods)
    X, W, Y = _rn_data(rng, n_sites, 10, share, n_periods, r_hi)
    od = oracle.OracleData(X, W, Y, model="occu_rn")
    ds = OccuDataset(X, W, Y, model="occu_rn")
    th = np.concatenate([rng.uniform(-1.0, 1.0, size=(3, od.D)), [[0.3, 0.2, -0.1, 2.5, 0.5, -0.5]], [[1.5, 0.1, 0.1, -1.0, 0.2, 0.2]]]).astype(np.float32).astype(np.float64)
    Uo, Go = od.potential_grad(th)
    Ug, Gg = ds.logp_grad(th)
    assert np.all(np.isfinite(Ug)) and np.all(np.isfinite(Gg))
    assert np.max(np.abs(Ug - Uo) / np.abs(Uo)) <= 1e-5, (Ug, Uo)
    assert np.max(np.abs(Gg - Go) / np.abs(Go).max(1, keepdims=True)) <= 1e-4, np.abs(Gg - Go).max(1) / np.abs(Go).max(1)
    if n_sites <= 5000:
        o = oracle.nuts_run(od, 0, 3, num_chains=2, seed=5)
        r = ds.nuts(num_warmup=0, num_samples=3, num_chains=2, seed=5, wgs_per_chain=k)
        assert np.array_equal(o["num_steps"][:, :2], r.num_steps[:, :2]), (o["num_steps"], r.num_steps)
        assert np.allclose(o["draws"][:, 0], r.draws[:, 0], atol=3e-3)
    ds.close()
