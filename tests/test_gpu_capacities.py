"""Every padded covariate-capacity pair is its own kernel instantiation (register allocation, unrolling and record
strides differ).  The model-specific test files concentrate on the capacities their fixtures have; this one runs the
extra-coordinate and count models at the largest capacity they are built for, and at two lopsided ones: K1 parity
through the C-ABI against the oracle, and the same first NUTS trees."""
import numpy as np
import pytest

import oracle
from biolith_amd.engine import OccuDataset

pytestmark = pytest.mark.gpu


def _data(rng, ks, ko, counts):
    N, T, J = 333, 2, 5
    X = rng.normal(size=(N, ks)) * 0.4
    W = rng.normal(size=(N, T, J, ko)) * 0.4
    if counts:
        Nn = rng.poisson(2.0, size=(N, T, 1))
        Y = rng.binomial(Nn, 0.45, size=(N, T, J)).astype(float)[None]
    else:
        Y = (rng.uniform(size=(1, N, T, J)) < 0.3) * 1.0
    Y[rng.uniform(size=Y.shape) < 0.08] = np.nan
    return X, W, Y


# (6, 9) runs the (8, 16) instantiation, (16, 16) the fullest one: the models that round 1 capped at 4 covariates per side
# (regression/linear.py:28-66 takes any n_covs)
@pytest.mark.parametrize("ks,ko", [(4, 4), (3, 1), (1, 3), (6, 9), (16, 16)])
@pytest.mark.parametrize("model", ["occu_fp", "occu_cop", "nmixture", "occu_rn"])
def test_model_capacities(model, ks, ko):
    rng = np.random.default_rng(100 * ks + 10 * ko + len(model))
    X, W, Y = _data(rng, ks, ko, counts=model in ("occu_cop", "nmixture"))
    if ks + ko > 8:   # many covariates: keep the linear predictors in a sane range
        X, W = X * 0.5, W * 0.5
    kw = dict(model=model)
    if model == "occu_fp":
        kw.update(fp_mode="unoccupied", prior_fp=(2.0, 6.0))
    elif model == "occu_cop":
        kw.update(fp_mode="constant", prior_fp_rate=1.5, session_duration=rng.uniform(1.0, 8.0, size=Y.shape[1:]))
    elif model == "occu_rn":
        kw.update(max_abundance=40)
    else:
        kw.update(max_abundance=int(np.nanmax(Y)) + 12)
    od, ds = oracle.OracleData(X, W, Y, **kw), OccuDataset(X, W, Y, **kw)
    assert ds.D == od.D
    th = rng.uniform(-0.7, 0.7, size=(3, od.D)).astype(np.float32).astype(np.float64)
    Uo, Go = od.potential_grad(th)
    Ug, Gg = ds.logp_grad(th)
    tol = 5.0 if model == "occu_rn" else 1.0   # (occu_rn's stated K1 tolerance is 1e-5 / 1e-4)
    assert np.max(np.abs(Ug - Uo) / np.abs(Uo)) <= 2e-6 * tol, (Ug, Uo)
    assert np.max(np.abs(Gg - Go)) <= 2e-5 * tol * np.max(np.abs(Go))
    o = oracle.nuts_run(od, 0, 4, num_chains=2, seed=3)
    r = ds.nuts(num_warmup=0, num_samples=4, num_chains=2, seed=3)
    assert np.array_equal(o["num_steps"][:, :3], r.num_steps[:, :3]), (o["num_steps"], r.num_steps)
    assert np.allclose(o["draws"][:, 0], r.draws[:, 0], atol=2e-3)


@pytest.mark.parametrize("ks,ko", [(6, 9), (16, 16)])
@pytest.mark.parametrize("model", ["occu_re", "occu_cs"])
def test_vector_kernel_capacities(model, ks, ko):
    """The random-effects / continuous-score kernels (re_kernel.hpp) at their 16-covariate instantiation."""
    rng = np.random.default_rng(7 * ks + ko)
    N, T, J = 120, 2, 4
    X = rng.normal(size=(N, ks)) * 0.2
    W = rng.normal(size=(N, T, J, ko)) * 0.2
    if model == "occu_cs":
        Y = rng.normal(size=(1, N, T, J)) + (rng.uniform(size=(1, N, T, J)) < 0.3) * 2.0
        kw = dict(model="occu_cs")
    else:
        Y = (rng.uniform(size=(1, N, T, J)) < 0.3) * 1.0
        kw = dict(model="occu_re", site_random_effects=True, obs_random_effects=True)
    Y[rng.uniform(size=Y.shape) < 0.08] = np.nan
    od, ds = oracle.OracleData(X, W, Y, **kw), OccuDataset(X, W, Y, **kw)
    assert ds.D == od.D
    th = rng.uniform(-0.5, 0.5, size=(2, od.D)).astype(np.float32).astype(np.float64)
    Uo, Go = od.potential_grad(th)
    Ug, Gg = ds.logp_grad(th)
    assert np.max(np.abs(Ug - Uo) / np.abs(Uo)) <= 4e-6, (Ug, Uo)
    assert np.max(np.abs(Gg - Go)) <= 4e-5 * np.max(np.abs(Go))
    # (a start near the bulk: from init_to_uniform's corners the float32 / float64 trajectories of this many coordinates part at once)
    init = rng.uniform(-0.3, 0.3, size=(2, od.D)).astype(np.float32).astype(np.float64)
    o = oracle.nuts_run(od, 0, 4, num_chains=2, seed=3, init=init)
    for k in (1, 3):
        r = ds.nuts(num_warmup=0, num_samples=4, num_chains=2, seed=3, wgs_per_chain=k, init_theta=init)
        assert np.array_equal(o["num_steps"][:, :3], r.num_steps[:, :3]), (k, o["num_steps"], r.num_steps)
        assert np.allclose(o["draws"][:, 0], r.draws[:, 0], atol=3e-3)
