"""NUTS against NUMERICAL INTEGRATION of the posterior (tests/quadrature.py): an anchor for the sampler that is external to both
restatements of numpyro's algorithm -- neither the oracle nor the engine wrote the answer.  Here: the oracle's NUTS (CPU).
What this pins: the whole chain density -> leapfrog -> tree -> multinomial -> adaptation SAMPLES THE RIGHT DISTRIBUTION (means within
4 Monte-Carlo standard errors, standard deviations within 3 %, correlations, five points of a marginal CDF).  What it does not pin:
that numpyro would build the same trees."""
import numpy as np
import pytest

import oracle
import quadrature as Q


@pytest.mark.parametrize("ks,ppa,seed", [(0, 301, 0), (1, 91, 1)])
def test_oracle_nuts_samples_the_integrated_posterior(ks, ppa, seed):
    X, W, Y = Q.tiny_occupancy_data(ks=ks, seed=seed)
    od = oracle.OracleData(X, W, Y)
    q = Q.grid_posterior(od, ppa)
    r = oracle.nuts_run(od, 500, 4000, num_chains=4, seed=11)
    Q.check_draws(r["draws"], q, oracle.effective_sample_size)
    assert oracle.split_gelman_rubin(r["draws"]).max() < 1.01
