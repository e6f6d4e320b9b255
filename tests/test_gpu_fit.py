"""fit(occu, ...) end to end on the MI355X, written like the reference's own inline tests
(biolith/models/occu.py:433-492) with the reference's tolerances."""
import time

import numpy as np
import pytest

from biolith_amd.evaluation import diagnostics
from biolith_amd.models import occu, simulate
from biolith_amd.utils import fit
from biolith_amd.utils.misc import TimeoutException
from biolith_amd import _ffi

pytestmark = pytest.mark.gpu


def test_engine_is_the_native_library():
    assert _ffi.device_count() >= 1
    assert _ffi.load().bl_abi_version() == 1


def test_occu():  # occu.py:433-456
    data, true_params = simulate(simulate_missing=True)
    results = fit(occu, **data, timeout=600)
    assert np.allclose(results.samples["psi"].mean(), true_params["z"].mean(), atol=0.1)
    assert np.allclose(
        [results.samples[k].mean() for k in [f"cov_state_{i}" for i in range(true_params["beta"].shape[1])]],
        true_params["beta"].mean(axis=0), atol=0.5)
    assert np.allclose(
        [results.samples[k].mean() for k in [f"cov_det_{i}" for i in range(true_params["alpha"].shape[1])]],
        true_params["alpha"].mean(axis=0), atol=0.5)
    # sample-dict contract: defaults are 5 chains x 1000 draws
    s = results.samples
    assert s["cov_state_0"].shape == (5000, 1) and s["cov_det_1"].shape == (5000, 1)
    assert s["psi"].shape == (5000, 1, 100, 1)
    assert s["prob_detection"].shape == (5000, 52, 1, 100, 1)
    assert "beta" not in s and "alpha" not in s
    # occu.py:229-235 under enumeration: the z axis (0 = unoccupied, 1 = occupied) in front of the plates; lazy
    pfp = s["prob_detection_fp"]
    assert pfp.shape == (5000, 2, 52, 1, 100, 1)
    assert np.all(pfp[:, 0] == 0.0) and np.array_equal(pfp[:, 1], s["prob_detection"])   # no false positives: z p
    d = diagnostics(results.mcmc)
    assert d["mean_r_hat"] < 1.01 and d["mean_frac_eff"] > 0.3 and d["frac_diverging"] < 0.01
    assert results.mcmc.num_chains == 5 and results.mcmc.num_samples == 1000
    assert results.mcmc.get_samples(group_by_chain=True)["beta"].shape == (5, 1000, 1, 2)


def test_occu_multi_season():  # occu.py:459-475
    data, true_params = simulate(simulate_missing=True, n_periods=3)
    results = fit(occu, **data, num_chains=1, num_samples=300, num_warmup=300, timeout=600)
    assert results.samples["psi"].shape == (300, 3, 100, 1)
    assert np.allclose(results.samples["psi"].mean(), true_params["z"].mean(), atol=0.15)


def test_occu_multi_species():  # occu.py:478-492
    data, _ = simulate(simulate_missing=True, n_species=2, n_sites=30)
    results = fit(occu, **data, num_chains=1, num_samples=100, num_warmup=100, timeout=600)
    assert results.samples["psi"].shape[-1] == 2
    assert results.samples["psi"].shape == (100, 1, 30, 2)
    assert results.samples["cov_state_0"].shape == (100, 2) and results.samples["prob_detection"].shape == (100, 52, 1, 30, 2)
    assert not np.allclose(results.samples["cov_det_0"][:, 0], results.samples["cov_det_0"][:, 1])
    # the species-by-species form (joint_species=False): species are independent given the shared covariates, so species 0
    # alone gives the same draws (the default is ONE chain over all species, as in the reference: tests/test_gpu_species.py)
    sep = fit(occu, **data, num_chains=1, num_samples=100, num_warmup=100, joint_species=False)
    one = dict(data, obs=data["obs"][:1])
    r1 = fit(occu, **one, num_chains=1, num_samples=100, num_warmup=100)
    assert np.array_equal(r1.samples["cov_state_0"][:, 0], sep.samples["cov_state_0"][:, 0])


def test_fit_is_seeded():
    data, _ = simulate(n_sites=50, random_seed=3)
    a = fit(occu, **data, num_chains=2, num_samples=50, num_warmup=50, random_seed=1)
    b = fit(occu, **data, num_chains=2, num_samples=50, num_warmup=50, random_seed=1)
    c = fit(occu, **data, num_chains=2, num_samples=50, num_warmup=50, random_seed=2)
    assert np.array_equal(a.samples["cov_state_0"], b.samples["cov_state_0"])
    assert not np.array_equal(a.samples["cov_state_0"], c.samples["cov_state_0"])


def test_dataframe_inputs_name_the_coefficients():
    import pandas as pd
    data, _ = simulate(n_sites=40, n_site_covs=2, n_obs_covs=1, deployment_days_per_site=35, random_seed=5)
    site_df = pd.DataFrame(data["site_covs"], columns=["elev", "forest"])
    r = fit(occu, site_covs=site_df, obs_covs=data["obs_covs"], obs=data["obs"], num_chains=1, num_samples=40, num_warmup=40)
    assert {"cov_state_intercept", "cov_state_elev", "cov_state_forest", "cov_det_0", "cov_det_1"} <= set(r.samples)


def test_timeout_raises_like_reference():  # fit.py:124-128 -> misc.py:11-21
    data, _ = simulate(n_sites=5000, n_site_covs=2, n_obs_covs=2, deployment_days_per_site=70, random_seed=1)
    t0 = time.time()
    with pytest.raises((TimeoutException, TimeoutError)):
        fit(occu, **data, num_chains=4, num_samples=500000, num_warmup=500000, timeout=1)
    assert time.time() - t0 < 10


def test_chains_dealt_over_devices_reproduce_the_single_launch():  # fit.py:109-113 chain_method="parallel"
    # the same GPU named twice stands in for two GPUs: chains 0-1 and 2-3 run as two launches, and each
    # chain's RNG streams depend on its global id only, so the draws equal those of one 4-chain launch
    # (the workgroup count per chain is pinned by the data size, not the chain count)
    data, _ = simulate(n_sites=300, n_site_covs=2, n_obs_covs=2, deployment_days_per_site=35, random_seed=2)
    kw = dict(num_chains=4, num_samples=60, num_warmup=60, random_seed=9)
    one = fit(occu, **data, **kw)
    two = fit(occu, **data, **kw, devices=[0, 0])
    three = fit(occu, **data, **kw, devices=[0, 0, 0])  # ragged deal: 2 + 1 + 1
    assert set(one.samples) == set(two.samples) == set(three.samples)
    for k in one.samples:
        assert np.array_equal(one.samples[k], two.samples[k]), k
        assert np.array_equal(one.samples[k], three.samples[k]), k
    assert np.array_equal(one.mcmc.get_extra_fields()["diverging"], two.mcmc.get_extra_fields()["diverging"])
    assert two.mcmc.num_chains == 4
    with pytest.raises(ValueError):
        fit(occu, **data, **kw, devices=[])


def test_init_strategies():  # fit.py:29, 93: NUTS(model_fn, init_strategy=init_strategy or init_to_uniform)
    from biolith_amd.utils import init_to_feasible, init_to_median, init_to_uniform, init_to_value

    data, truth = simulate(n_sites=300, n_site_covs=2, n_obs_covs=2, deployment_days_per_site=70, random_seed=3)
    kw = dict(num_chains=3, num_warmup=0, num_samples=4, random_seed=1)
    base = fit(occu, **data, **kw)
    same = fit(occu, **data, **kw, init_strategy=init_to_uniform())                 # the default by its name: the kernel's own draw
    assert np.array_equal(base.mcmc.result.draws, same.mcmc.result.draws)
    other = fit(occu, **data, **kw, init_strategy=init_to_uniform(radius=0.1))
    assert not np.array_equal(base.mcmc.result.draws, other.mcmc.result.draws)
    # no warmup, step size 1: from the generating values the first draws stay in their neighbourhood; from Uniform(-2, 2) they do not
    start = init_to_value(values={"beta": truth["beta"], "alpha": truth["alpha"]})
    near = fit(occu, **data, **{**kw, "num_samples": 1}, init_strategy=start)
    th = np.concatenate([truth["beta"][0], truth["alpha"][0]])
    assert np.max(np.abs(near.mcmc.result.draws[:, 0] - th)) < 1.0
    # the dealing of chains to launches does not change where a chain starts
    split = fit(occu, **data, **kw, init_strategy=init_to_uniform(radius=0.1), devices=[0, 0])
    assert np.array_equal(split.mcmc.result.draws, other.mcmc.result.draws)
    for strat in (init_to_feasible(), init_to_median(num_samples=5)):
        res = fit(occu, **data, num_chains=2, num_warmup=200, num_samples=200, init_strategy=strat)
        assert abs(float(res.samples["psi"].mean()) - truth["z"].mean()) < 0.1      # occu.py:440
    with pytest.raises(NotImplementedError, match="all regression coefficients"):
        fit(occu, **data, **kw, false_positives_constant=True, init_strategy=init_to_median())
