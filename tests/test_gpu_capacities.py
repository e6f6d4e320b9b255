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


@pytest.mark.parametrize("ks,ko", [(4, 4), (3, 1), (1, 3)])
@pytest.mark.parametrize("model", ["occu_fp", "occu_cop", "nmixture"])
def test_model_capacities(model, ks, ko):
    rng = np.random.default_rng(100 * ks + 10 * ko + len(model))
    X, W, Y = _data(rng, ks, ko, counts=model != "occu_fp")
    kw = dict(model=model)
    if model == "occu_fp":
        kw.update(fp_mode="unoccupied", prior_fp=(2.0, 6.0))
    elif model == "occu_cop":
        kw.update(fp_mode="constant", prior_fp_rate=1.5, session_duration=rng.uniform(1.0, 8.0, size=Y.shape[1:]))
    else:
        kw.update(max_abundance=int(np.nanmax(Y)) + 12)
    od, ds = oracle.OracleData(X, W, Y, **kw), OccuDataset(X, W, Y, **kw)
    assert ds.D == od.D
    th = rng.uniform(-0.7, 0.7, size=(3, od.D)).astype(np.float32).astype(np.float64)
    Uo, Go = od.potential_grad(th)
    Ug, Gg = ds.logp_grad(th)
    assert np.max(np.abs(Ug - Uo) / np.abs(Uo)) <= 2e-6, (Ug, Uo)
    assert np.max(np.abs(Gg - Go)) <= 2e-5 * np.max(np.abs(Go))
    o = oracle.nuts_run(od, 0, 4, num_chains=2, seed=3)
    r = ds.nuts(num_warmup=0, num_samples=4, num_chains=2, seed=3)
    assert np.array_equal(o["num_steps"][:, :3], r.num_steps[:, :3]), (o["num_steps"], r.num_steps)
    assert np.allclose(o["draws"][:, 0], r.draws[:, 0], atol=2e-3)
