#!/usr/bin/env python3
"""Is the random-effects scale's SBC failure the funnel or the kernel?  (models/occu.py:170-173, 191-196, 215-218)

Simulation-based calibration of occu with site or observation random effects (tests/sbc.py: parameters from the prior, data from the
generative model, rank of the truth among thinned draws) at several `target_accept` and thinnings.  A funnel bias -- NUTS with ONE step
size cannot enter the neck where sd is small -- shrinks as target_accept -> 0.99 (smaller steps reach further down) and is untouched by
thinning; an implementation bias (a wrong Jacobian, a wrong half step) does neither.

    python tools/sbc_re_sweep.py --backend engine --seeds 5 --accept 0.8 0.95 0.99 --thin 5 20 > profiles/r06/...
    python tools/sbc_re_sweep.py --backend oracle --seeds 5 --accept 0.8 0.99 --thin 5 --effects obs     (CPU: the float64 restatement)

One line per (effects, seed, target_accept, thin): chi-square of the ranks (10 bins; limit = 99.9 % point of chi2(9) = 27.9) for the four
coefficients, log sd and the first four effects; the ten bin counts of log sd; divergences; mean leapfrogs per transition."""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import sbc  # noqa: E402


def run(backend, site_re, obs_re, seed, accept, thin, reps, warmup, n_sites, n_visits, noncentred_check=False):
    rng = np.random.default_rng(seed + 2)     # (tests/test_gpu_sbc.py's stream: SEED + 2)
    samples = 50 * thin                       # 4 chains x samples / thin = 200 kept, the first 199 used
    ranks, div, steps, M = [], 0, [], None
    t0 = time.time()
    for l in range(reps):
        X, W, Y, theta, kw = sbc.prior_predictive_re(rng, n_sites, n_visits, 1, 1, site_re, obs_re)
        if backend == "engine":
            from biolith_amd.engine import OccuDataset

            ds = OccuDataset(X, W, Y, **kw)
            r = ds.nuts(num_warmup=warmup, num_samples=samples, num_chains=4, seed=l, target_accept=accept)
            ds.close()
            draws, dv, ns = r.draws, int(r.diverging.sum()), float(r.num_steps.mean())
        else:
            import oracle

            od = oracle.OracleData(X, W, Y, **kw)
            r = oracle.nuts_run(od, warmup, samples, num_chains=4, seed=l, target_accept=accept)
            draws, dv, ns = r["draws"], int(r["diverging"].sum()), float(r["num_steps"].mean())
        div += dv
        steps.append(ns)
        rk, M = sbc.rank_of_truth(np.asarray(draws, dtype=np.float64), theta, thin, 199)
        ranks.append(rk)
    ranks = np.stack(ranks)[:, :9]
    stat, crit, counts = sbc.uniformity(ranks, M, bins=10)
    return dict(backend=backend, effects="site" if site_re else "obs", seed=seed, target_accept=accept, thin=thin, reps=reps,
                warmup=warmup, samples=samples, chi2=[round(float(x), 1) for x in stat], chi2_limit=round(crit, 1),
                chi2_log_sd=round(float(stat[4]), 1), log_sd_ok=bool(stat[4] < crit), coefficients_ok=bool(np.all(stat[:4] < crit)),
                log_sd_bins=counts[4].tolist(), divergences=div, mean_num_steps=round(float(np.mean(steps)), 1),
                seconds=round(time.time() - t0, 1))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--backend", choices=("engine", "oracle"), default="engine")
    ap.add_argument("--effects", nargs="+", choices=("site", "obs"), default=["site", "obs"])
    ap.add_argument("--seeds", nargs="+", type=int, default=[5])
    ap.add_argument("--accept", nargs="+", type=float, default=[0.8, 0.95, 0.99])
    ap.add_argument("--thin", nargs="+", type=int, default=[5, 20])
    ap.add_argument("--reps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=500)
    ap.add_argument("--sites", type=int, default=40)
    ap.add_argument("--visits", type=int, default=6)
    a = ap.parse_args()
    for eff in a.effects:
        for seed in a.seeds:
            for acc in a.accept:
                for thin in a.thin:
                    row = run(a.backend, eff == "site", eff == "obs", seed, acc, thin, a.reps, a.warmup, a.sites, a.visits)
                    print(json.dumps(row), flush=True)


if __name__ == "__main__":
    main()
