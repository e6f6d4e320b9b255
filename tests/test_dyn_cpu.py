"""Dynamic occupancy (BUILDER-DEFINED, no reference counterpart: BASELINE.json configs[4] names a model timmh/biolith does not
have) -- what pins the builder's own oracle: the C forward / backward recursions against a literal statement of the model with
the 2^T latent paths summed by brute force, and the analytic gradient against central differences."""
import contextlib
import io

import numpy as np
import pytest

import oracle
from biolith_amd.models import simulate_dyn


def _data(rng, N=6, T=4, J=3, Ks=2, Ko=2, missing=True):
    X = rng.normal(size=(N, Ks))
    W = rng.normal(size=(N, T, J, Ko))
    Y = (rng.uniform(size=(N, T, J)) < 0.4) * 1.0
    if missing:
        Y[1, 2, 1] = np.nan
        Y[2, 0, :] = np.nan          # a whole season unobserved: the recursion just propagates the state
        W[3, 1, 0, 1] = np.nan
        X[4, 0] = np.nan             # a site covariate missing: every visit of the site is masked (occu.py:136-142)
    return X, W, Y


@pytest.mark.parametrize("T", [1, 2, 5])
def test_forward_recursion_equals_brute_force_over_paths(T):
    rng = np.random.default_rng(T)
    X, W, Y = _data(rng, T=max(T, 3))
    W, Y = W[:, :T], Y[:, :T]
    od = oracle.OracleData(X, W, Y, (0.2, 1.5), (-0.1, 0.8), model="occu_dyn")
    assert od.D == 3 * 3 + 3
    for _ in range(3):
        th = rng.uniform(-2, 2, size=od.D)
        U, _ = od.potential_grad(th)
        assert abs(U + oracle.literal_log_joint_dyn(th, X, W, Y, (0.2, 1.5), (-0.1, 0.8))) < 1e-10 * max(1.0, abs(U))


def test_gradient_equals_central_differences():
    rng = np.random.default_rng(3)
    X, W, Y = _data(rng, N=40, T=6, J=4)
    od = oracle.OracleData(X, W, Y, model="occu_dyn")
    for _ in range(3):
        th = rng.uniform(-1.5, 1.5, size=od.D)
        _, G = od.potential_grad(th)
        h = 1e-6
        fd = np.array([(od.potential_grad(th + h * e)[0] - od.potential_grad(th - h * e)[0]) / (2 * h) for e in np.eye(od.D)])
        assert np.max(np.abs(fd - G)) <= 1e-6 * max(1.0, np.max(np.abs(G)))


def test_one_season_is_the_plain_occupancy_model():
    """T = 1: no transition is ever taken, so the density is occu's (the colonisation / extinction blocks only see their priors)."""
    rng = np.random.default_rng(5)
    X, W, Y = _data(rng, T=3)
    W, Y = W[:, :1], Y[:, :1]
    dyn, occ = oracle.OracleData(X, W, Y, model="occu_dyn"), oracle.OracleData(X, W, Y)
    th = rng.uniform(-1, 1, size=dyn.D)
    B = X.shape[1] + 1
    th_occ = np.concatenate([th[:B], th[3 * B:]])
    prior_rest = 0.5 * np.sum(th[B:3 * B] ** 2) + 2 * B * 0.5 * np.log(2 * np.pi)
    Ud, Gd = dyn.potential_grad(th)
    Uo, Go = occ.potential_grad(th_occ)
    assert abs(Ud - (Uo + prior_rest)) < 1e-10
    assert np.allclose(np.concatenate([Gd[:B], Gd[3 * B:]]), Go, atol=1e-10) and np.allclose(Gd[B:3 * B], th[B:3 * B], atol=1e-12)


def test_oracle_sampler_recovers_the_generating_rates():
    with contextlib.redirect_stdout(io.StringIO()):
        data, truth = simulate_dyn(n_sites=300, n_periods=6, random_seed=1)
    od = oracle.OracleData(data["site_covs"], data["obs_covs"], data["obs"], model="occu_dyn")
    r = oracle.nuts_run(od, 300, 300, num_chains=2, seed=0)
    assert r["diverging"].sum() == 0 and oracle.split_gelman_rubin(r["draws"]).max() < 1.05
    flat = r["draws"].reshape(-1, od.D)
    X = data["site_covs"].astype(np.float32).astype(np.float64)
    sig = lambda v: 1 / (1 + np.exp(-v))  # noqa: E731
    for b, name in enumerate(("psi", "gamma", "epsilon")):
        est = sig(flat[:, 2 * b:2 * b + 1] + flat[:, 2 * b + 1:2 * b + 2] @ X.T).mean()
        assert abs(est - truth[name].mean()) < 0.1, (name, est, truth[name].mean())


def test_simulate_dyn_shapes_and_missingness():
    with contextlib.redirect_stdout(io.StringIO()):
        data, truth = simulate_dyn(n_sites=50, n_periods=4, n_site_covs=2, n_obs_covs=3, simulate_missing=True, random_seed=2)
    assert data["site_covs"].shape == (50, 2) and data["obs_covs"].shape == (50, 4, 4, 3) and data["obs"].shape == (1, 50, 4, 4)
    assert np.isnan(data["obs"]).mean() > 0.1 and truth["z"].shape == (4, 50)
    det = np.nan_to_num(data["obs"][0])                       # no detection where the site is unoccupied that season
    assert np.all(det[truth["z"].T == 0] == 0)
