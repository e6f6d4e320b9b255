"""Randomised shapes through every model of the engine: potential and gradient over all coordinates against the CPU oracle.
Sixty seeded draws of (model, sites, periods, visits, covariate counts, missing-data pattern, priors) -- the ragged corners the
hand-written cases do not name: one site, one visit, no covariates on one side or both, sites or periods without data, slices that
do not fill a workgroup, 5-16 covariates (the capacity-8 / 16 kernels)."""
import numpy as np
import pytest

import oracle
from biolith_amd.engine import OccuDataset

pytestmark = pytest.mark.gpu

MODELS = ["occu", "occu_fp_c", "occu_fp_u", "occu_rn", "occu_cop", "occu_cop_fp", "nmixture", "occu_re_s", "occu_re_o", "occu_re_so", "occu_cs"]
JOINT = ["occu", "occu_fp_c", "occu_fp_u", "occu_re_s", "occu_re_so"]   # several species under one chain


def _draw(seed):
    rng = np.random.default_rng(1000 + seed)
    model = MODELS[seed % len(MODELS)]
    N = int(rng.choice([1, 2, 3, 17, 64, 65, 130, 333, 700]))
    T = int(rng.choice([1, 1, 2, 3]))
    J = int(rng.choice([1, 2, 5, 9]))
    Ks = int(rng.choice([0, 1, 3, 4, 5, 8, 11, 16]))
    Ko = int(rng.choice([0, 1, 2, 4, 6, 8, 16]))
    X = rng.normal(size=(N, Ks)).astype(np.float32) * 0.7
    W = rng.normal(size=(N, T, J, Ko)).astype(np.float32) * 0.7
    kw = {}
    if model in ("occu_cop", "occu_cop_fp"):
        Y = rng.poisson(0.8, size=(1, N, T, J)).astype(np.float32)
        kw = dict(model="occu_cop", session_duration=rng.uniform(0.5, 3.0, size=(N, T, J)).astype(np.float32),
                  fp_mode="constant" if model.endswith("fp") else None)
    elif model == "nmixture":
        Y = rng.binomial(3, 0.4, size=(1, N, T, J)).astype(np.float32)
        kw = dict(model="nmixture", max_abundance=int(rng.choice([6, 20, 60])))
    elif model == "occu_cs":
        Y = rng.normal(1.0, 1.5, size=(1, N, T, J)).astype(np.float32)
        kw = dict(model="occu_cs", prior_mu=((0.5, 8.0), (1.0, 12.0)), prior_sigma=((5.0, 1.0), (3.0, 0.5)))
    else:
        Y = (rng.uniform(size=(1, N, T, J)) < 0.35).astype(np.float32)
        if model.startswith("occu_fp"):
            kw = dict(model="occu_fp", fp_mode="constant" if model.endswith("c") else "unoccupied", prior_fp=(2.0, 6.0))
        elif model == "occu_rn":
            kw = dict(model="occu_rn", max_abundance=int(rng.choice([5, 30, 100])))
        elif model.startswith("occu_re"):
            kw = dict(model="occu_re", site_random_effects="s" in model.split("_")[2], obs_random_effects="o" in model.split("_")[2],
                      prior_site_re_sd=0.8, prior_obs_re_sd=1.2)
    # missing data: single visits, a whole period, a whole site, NaN covariates
    pattern = int(rng.integers(0, 5))
    if pattern >= 1:
        Y[0][rng.uniform(size=(N, T, J)) < 0.15] = np.nan
    if pattern >= 2 and N > 1:
        Y[0, int(rng.integers(0, N))] = np.nan
    if pattern >= 3 and T > 1:
        Y[0, :, int(rng.integers(0, T))] = np.nan
    if pattern >= 4 and Ko > 0:
        W[int(rng.integers(0, N)), 0, int(rng.integers(0, J)), 0] = np.nan
    if pattern >= 4 and Ks > 0 and N > 2:
        X[int(rng.integers(0, N)), 0] = np.nan
    priors = ((float(rng.normal()) * 0.3, float(rng.uniform(0.5, 2.0))), (float(rng.normal()) * 0.3, float(rng.uniform(0.5, 2.0))))
    return rng, model, X, W, Y, priors, kw


@pytest.mark.parametrize("seed", range(66))
def test_random_shape_potential_and_gradient(seed):
    rng, model, X, W, Y, priors, kw = _draw(seed)
    od = oracle.OracleData(X, W, Y[0], *priors, **kw)
    ds = OccuDataset(X, W, Y, *priors, **kw)
    assert od.D == ds.D, (model, od.D, ds.D)
    th = rng.uniform(-1.0, 1.0, size=(2, od.D))
    if kw.get("model") == "occu_cs":
        th[:, -4:] = np.array([0.3, 1.2, 0.4, 0.2]) + rng.uniform(-0.2, 0.2, size=(2, 4))
    th = th.astype(np.float32).astype(np.float64)
    Uo, Go = od.potential_grad(th)
    Ug, Gg = ds.logp_grad(th)
    assert np.all(np.isfinite(Ug)) and np.all(np.isfinite(Gg)), model
    scale = np.maximum(np.abs(Uo), 1.0)
    assert np.max(np.abs(Ug - Uo) / scale) <= 3e-6, (model, X.shape, W.shape, Ug, Uo)
    assert np.max(np.abs(Gg - Go)) <= 5e-5 * max(np.max(np.abs(Go)), 1.0), (model, X.shape, W.shape, np.max(np.abs(Gg - Go)), np.max(np.abs(Go)))


@pytest.mark.parametrize("seed", range(33))
def test_random_shape_first_trees(seed):
    """The same random shapes through the sampler: the first two trees of two chains equal the oracle's (same xoshiro streams),
    the first draws agree to float32 accuracy; on one workgroup per chain and on the engine's own choice."""
    rng, model, X, W, Y, priors, kw = _draw(seed)
    if X.shape[0] > 333:
        X, W, Y = X[:333], W[:333], Y[:, :333]
        if "session_duration" in kw:
            kw["session_duration"] = kw["session_duration"][:333]
    od, ds = oracle.OracleData(X, W, Y[0], *priors, **kw), OccuDataset(X, W, Y, *priors, **kw)
    init = rng.uniform(-0.3, 0.3, size=(2, od.D))
    if kw.get("model") == "occu_cs":
        init[:, -4:] = np.array([0.3, 1.2, 0.4, 0.2])
    init = init.astype(np.float32).astype(np.float64)
    o = oracle.nuts_run(od, 0, 3, num_chains=2, seed=seed, init=init)
    for k in (1, 0):
        r = ds.nuts(num_warmup=0, num_samples=3, num_chains=2, seed=seed, init_theta=init, wgs_per_chain=k)
        # (a transition that diverges -- energy error beyond 1000 at step size 1 on a two-site dataset, say -- may stop at another
        # leaf in float32 than in float64: trees are compared up to the first divergence of a chain)
        ok = np.logical_and.accumulate(~(np.asarray(o["diverging"][:, :2], bool) | np.asarray(r.diverging[:, :2], bool)), axis=1)
        assert np.array_equal(o["num_steps"][:, :2][ok], r.num_steps[:, :2][ok]), (model, k, o["num_steps"], r.num_steps)
        assert np.allclose(o["draws"][:, 0][ok[:, 0]], r.draws[:, 0][ok[:, 0]], atol=5e-3), (model, k)


@pytest.mark.parametrize("seed", range(20))
def test_random_shape_several_species(seed):
    """Two or three species under one chain (shared false-positive rate / shared sds): potential and gradient over all coordinates."""
    rng = np.random.default_rng(5000 + seed)
    model = JOINT[seed % len(JOINT)]
    S, N, T, J = int(rng.choice([2, 3])), int(rng.choice([2, 40, 129, 300])), int(rng.choice([1, 2])), int(rng.choice([1, 4, 7]))
    Ks, Ko = int(rng.choice([0, 1, 3, 4])), int(rng.choice([0, 2, 4]))
    X = rng.normal(size=(N, Ks)).astype(np.float32) * 0.7
    W = rng.normal(size=(N, T, J, Ko)).astype(np.float32) * 0.7
    Y = (rng.uniform(size=(S, N, T, J)) < 0.35).astype(np.float32)
    Y[rng.uniform(size=Y.shape) < 0.1] = np.nan
    if Ko:
        W[int(rng.integers(0, N)), 0, int(rng.integers(0, J)), 0] = np.nan
    kw = {}
    if model.startswith("occu_fp"):
        kw = dict(model="occu_fp", fp_mode="constant" if model.endswith("c") else "unoccupied", prior_fp=(2.0, 6.0))
    elif model.startswith("occu_re"):
        kw = dict(model="occu_re", site_random_effects=True, obs_random_effects=model.endswith("so"), prior_site_re_sd=0.8, prior_obs_re_sd=1.2)
    od, ds = oracle.OracleData(X, W, Y, **kw), OccuDataset(X, W, Y, **kw)
    assert od.D == ds.D and od.n_species == S
    th = rng.uniform(-1.0, 1.0, size=(2, od.D)).astype(np.float32).astype(np.float64)
    Uo, Go = od.potential_grad(th)
    Ug, Gg = ds.logp_grad(th)
    assert np.max(np.abs(Ug - Uo) / np.maximum(np.abs(Uo), 1.0)) <= 3e-6, (model, S, X.shape, W.shape, Ug, Uo)
    assert np.max(np.abs(Gg - Go)) <= 5e-5 * max(np.max(np.abs(Go)), 1.0), (model, S, X.shape, W.shape)
    init = rng.uniform(-0.3, 0.3, size=(2, od.D)).astype(np.float32).astype(np.float64)
    o = oracle.nuts_run(od, 0, 3, num_chains=2, seed=seed, init=init)
    r = ds.nuts(num_warmup=0, num_samples=3, num_chains=2, seed=seed, init_theta=init)
    ok = np.logical_and.accumulate(~(np.asarray(o["diverging"][:, :2], bool) | np.asarray(r.diverging[:, :2], bool)), axis=1)
    assert np.array_equal(o["num_steps"][:, :2][ok], r.num_steps[:, :2][ok]), (model, o["num_steps"], r.num_steps)
