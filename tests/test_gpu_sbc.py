"""Simulation-based calibration of the HIP engine (tests/sbc.py; the CPU twin for the oracle is tests/test_sbc.py): parameters from the
prior, data from the reference's generative model, the posterior through the C-ABI, ranks of the truth among thinned draws -- uniform
for a correct sampler + density, whatever the data size.  No restatement of the density and no numpyro are involved: the engine is
held to the MODEL (biolith/models/occu.py:136-242, occu_rn.py:123-222)."""
import os

import numpy as np
import pytest

import sbc
from biolith_amd.engine import OccuDataset

pytestmark = pytest.mark.gpu

SEED = int(os.environ.get("SBC_SEED", "5"))   # (other seeds: profiles/r05/s_gpu_sbc_other_seeds.txt)

SHAPE = dict(n_sites=40, n_visits=4, ks=1, ko=1)


def _engine_posterior(model, k, divergences, kernels, warmup=300, samples=250):
    def post(X, W, Y, l, **kw):
        ds = OccuDataset(X, W, Y, **kw)
        r = ds.nuts(num_warmup=warmup, num_samples=samples, num_chains=4, seed=l, wgs_per_chain=k)
        ds.close()
        divergences.append(int(r.diverging.sum()))
        kernels.add(r.kernel_name.strip())
        return r.draws
    return post


@pytest.mark.parametrize("model,reps,k,shape", [
    ("occu", 400, 0, dict(SHAPE)),                                                  # one workgroup per chain
    ("occu", 300, 0, dict(SHAPE, n_periods=2, missing=0.3, ks=2)),
    ("occu", 200, 3, dict(SHAPE, n_sites=600, n_visits=5, ks=3, ko=3)),           # three workgroups per chain: the exchange
    ("occu", 200, 0, dict(SHAPE, n_sites=30, n_visits=40, ko=2)),                 # lane groups over the visits
    ("occu", 300, 0, dict(SHAPE, fp="constant", n_visits=6)),
    ("occu", 300, 0, dict(SHAPE, fp="unoccupied", n_visits=6)),
    ("occu_cop", 300, 0, dict(SHAPE)),
    ("occu_cop", 300, 0, dict(SHAPE, fp="constant")),
    ("occu_cop", 300, 0, dict(SHAPE, fp="unoccupied", n_periods=2)),
    ("occu", 100, 0, dict(SHAPE, n_sites=2000, n_periods=8, n_visits=4, ks=3, ko=3)),   # config 5's stand-in shape: one period per lane
    ("occu_rn", 200, 0, dict(SHAPE, n_sites=60)),
    ("occu_rn", 60, 0, dict(SHAPE, n_sites=1500, n_visits=10, ks=3, ko=3)),              # several workgroups per chain
    # (occu_rn with a false-positive probability -- the random-effects kernel's Royle-Nichols kind -- is calibrated too, at 200 replications and
    #  two seeds; at 6 minutes a run it is kept out of the suite: profiles/r05/s_gpu_sbc_rn_false_positives.txt, sbc.prior_predictive(fp="constant"))
    ("nmixture", 200, 0, dict(SHAPE, beta_scale=0.7)),   # (prior_beta = Normal(0, 0.7) on both sides: few replications reach the sum's bound K = 100 and are redrawn, tests/sbc.py)
    ("occu_dyn", 300, 0, dict(SHAPE, n_sites=80, n_periods=4, n_visits=3)),      # (no reference counterpart: the builder's model)
    ("occu_dyn", 200, 0, dict(SHAPE, n_sites=200, n_periods=8, n_visits=4, ks=0)),  # the two-scans form: one period per lane
])
def test_engine_ranks_are_uniform(model, reps, k, shape):
    div, kernels = [], set()
    ranks, M = sbc.run(_engine_posterior(model, k, div, kernels), model, reps, seed=SEED, thin=5, keep=199, **shape)
    print(model, shape, sorted(kernels))
    assert M == 199 and sum(div) <= 0.002 * reps * 1000, (sum(div), max(div))
    stat, crit, counts = sbc.uniformity(ranks, M, bins=10)
    assert np.all(stat < crit), (stat, crit, counts)
    L = ranks.shape[0]
    sd_u = np.sqrt(M * (M + 2) / 12.0)
    assert np.all(np.abs(ranks.mean(0) - M / 2.0) < 4.0 * sd_u / np.sqrt(L)), ranks.mean(0)
    assert np.all(np.abs(ranks.std(0) / sd_u - 1.0) < 4.0 / np.sqrt(2.0 * L) * 1.35), ranks.std(0) / sd_u


@pytest.mark.parametrize("site_re,obs_re", [(True, False), (False, True)])
def test_engine_ranks_are_uniform_with_random_effects(site_re, obs_re):
    """occu with random effects (the other kernel, re_kernel.hpp: D = 85 / 245 here).  The COEFFICIENTS' ranks are asserted.  The centred
    effects ~ Normal(0, sd) under sd ~ HalfNormal(1) are a funnel, and NUTS -- the oracle's on the CPU exactly like this one
    (profiles/NOTES.md, round 5) -- over-states log sd where the true sd is small: strongly with observation effects (ONE binary observation
    per effect: the lowest tenth of the ranks holds 50 - 55 of 200 at every seed), mildly with site effects (six visits per effect: within the
    limit at seeds 5 - 7, 42 of 200 at seed 8).  log sd and the effects are therefore reported, not asserted."""
    rng = np.random.default_rng(SEED + 2)
    ranks, div = [], 0
    for l in range(200):
        X, W, Y, theta, kw = sbc.prior_predictive_re(rng, 40, 6, 1, 1, site_re, obs_re)
        ds = OccuDataset(X, W, Y, **kw)
        r = ds.nuts(num_warmup=500, num_samples=250, num_chains=4, seed=l)
        ds.close()
        div += int(r.diverging.sum())
        rk, M = sbc.rank_of_truth(r.draws.astype(np.float64), theta, 5, 199)
        ranks.append(rk)
    ranks = np.stack(ranks)[:, :9]                          # beta (2), alpha (2), log sd, the first four effects
    stat, crit, counts = sbc.uniformity(ranks, M, bins=10)
    print("random effects", site_re, obs_re, "divergences", div, "chi2", stat.round(1).tolist(), "log sd bins", counts[4].tolist())
    assert np.all(stat[:4] < crit), (stat, crit, counts)
    assert div <= 0.002 * 200 * 1000
