"""The HIP NUTS engine against NUMERICAL INTEGRATION of the posterior (tests/quadrature.py; the CPU twin for the oracle's sampler is
tests/test_sampler_vs_quadrature.py): means within 4 Monte-Carlo standard errors, standard deviations within 3 %, correlations within
0.03, five points of a marginal CDF -- for two and three coefficients, through the C-ABI, one workgroup and several per chain."""
import numpy as np
import pytest

import oracle
import quadrature as Q
from biolith_amd.engine import OccuDataset
from biolith_amd.evaluation import effective_sample_size, split_gelman_rubin

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("ks,ppa,seed,n_sites,k", [(0, 301, 0, 80, 0), (1, 91, 1, 80, 0), (1, 61, 2, 700, 3)])   # (61 points per axis: the same moments to 8 digits as 91, a third of the oracle evaluations)
def test_engine_nuts_samples_the_integrated_posterior(ks, ppa, seed, n_sites, k):
    X, W, Y = Q.tiny_occupancy_data(n_sites=n_sites, ks=ks, seed=seed)
    q = Q.grid_posterior(oracle.OracleData(X, W, Y), ppa)
    ds = OccuDataset(X, W, Y)
    r = ds.nuts(num_warmup=1000, num_samples=5000, num_chains=4, seed=3, wgs_per_chain=k)
    Q.check_draws(r.draws, q, effective_sample_size)
    assert split_gelman_rubin(r.draws).max() < 1.01 and int(r.diverging.sum()) == 0
    ds.close()
