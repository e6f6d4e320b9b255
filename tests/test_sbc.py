"""Simulation-based calibration of the ORACLE's sampler + density (tests/sbc.py; the GPU twin is tests/test_gpu_sbc.py): parameters from the
prior, data from the reference's generative model, ranks of the truth among thinned posterior draws -- uniform for a correct pair,
whatever the data size.  Pins the oracle's NUTS and its marginal density to the MODEL rather than to a restatement of either.  A
negative control shows the statistic's power at this number of replications."""
import numpy as np
import pytest

import oracle
import sbc

SHAPE = dict(n_sites=40, n_visits=4, ks=1, ko=1)


def _oracle_posterior(model, warmup=200, samples=250):
    def post(X, W, Y, l, **kw):
        return oracle.nuts_run(oracle.OracleData(X, W, Y, **kw), warmup, samples, num_chains=4, seed=l)["draws"]
    return post


@pytest.mark.parametrize("model,reps,shape", [
    ("occu", 200, dict(SHAPE)),
    ("occu", 150, dict(SHAPE, n_periods=2, missing=0.3, ks=2)),
    ("occu", 150, dict(SHAPE, fp="unoccupied", n_visits=6)),
    ("occu_cop", 100, dict(SHAPE, fp="constant")),
    ("occu_rn", 50, dict(SHAPE, n_sites=30)),
    ("occu_dyn", 100, dict(SHAPE, n_sites=60, n_periods=3, n_visits=3)),
])
def test_oracle_ranks_are_uniform(model, reps, shape):
    ranks, M = sbc.run(_oracle_posterior(model), model, reps, seed=11, thin=5, keep=199, **shape)
    assert M == 199                                  # 4 chains x 50 kept draws less one: 200 rank values in ten groups of 20
    stat, crit, counts = sbc.uniformity(ranks, M, bins=10 if reps >= 100 else 5)
    assert np.all(stat < crit), (stat, crit, counts)
    # the ranks' mean and spread against the uniform law's (mean M / 2, sd sqrt(M (M + 2) / 12)), 4 standard errors
    L = ranks.shape[0]
    sd_u = np.sqrt(M * (M + 2) / 12.0)
    assert np.all(np.abs(ranks.mean(0) - M / 2.0) < 4.0 * sd_u / np.sqrt(L)), ranks.mean(0)
    assert np.all(np.abs(ranks.std(0) / sd_u - 1.0) < 4.0 / np.sqrt(2.0 * L) * 1.35), ranks.std(0) / sd_u  # (uniform: kurtosis 1.8)


def test_a_wrong_posterior_is_caught():
    """Power: the same replications with every posterior narrowed to 70 % of its spread (an over-confident sampler), and with every
    posterior shifted by half its standard deviation (a biased one), must fail the chi-square test in EVERY coordinate."""
    def narrowed(X, W, Y, l, **kw):
        d = _oracle_posterior("occu")(X, W, Y, l)
        m = d.reshape(-1, d.shape[-1]).mean(0)
        return m + 0.7 * (d - m)

    def shifted(X, W, Y, l, **kw):
        d = _oracle_posterior("occu")(X, W, Y, l)
        return d + 0.5 * d.reshape(-1, d.shape[-1]).std(0)

    for wrong, what in ((narrowed, "spread"), (shifted, "mean")):
        ranks, M = sbc.run(wrong, "occu", 200, seed=11, thin=5, keep=199, **SHAPE)
        stat, crit, _ = sbc.uniformity(ranks, M, bins=10)
        assert np.all(stat > crit), (what, stat, crit)
