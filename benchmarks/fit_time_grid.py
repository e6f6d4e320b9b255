#!/usr/bin/env python3
"""Wall time of the whole ``fit(occu, ...)`` call over the dataset grid of the reference's own benchmark.

Methodology of biolith/benchmarks/occu_spoccupancy.py:16-118,436-452 (its biolith leg; the R/spOccupancy leg has no
counterpart here): datasets i = 0..7 with ``n_sites = 100 * 2^i``, ``int(8 * 2^(i/2))`` visits, 2 site + 1 observation
covariate, ``simulate(random_seed=42 + i)``, one chain, 500 draws after 100 warmup transitions, the timer around the
whole ``fit()`` call.  The first row therefore includes loading the engine library (the reference's first row includes
its JIT compile).  Prints one JSON object; ``python benchmarks/fit_time_grid.py > profiles/r01/g_fit_time_grid.json``.
"""
import contextlib
import io
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from biolith_amd.evaluation import effective_sample_size  # noqa: E402
from biolith_amd.models import occu, simulate  # noqa: E402
from biolith_amd.utils import fit  # noqa: E402


def main(n_datasets=8, num_samples=500, num_warmup=100, base_n_sites=100, base_time_periods=8, scaling_factor=2, random_seed=42):
    rows = []
    for i in range(n_datasets):
        n_sites = int(base_n_sites * (scaling_factor ** i))
        time_periods = int(base_time_periods * (scaling_factor ** (i / 2)))
        with contextlib.redirect_stdout(io.StringIO()):
            data, truth = simulate(n_site_covs=2, n_obs_covs=1, n_sites=n_sites, deployment_days_per_site=time_periods * 7,
                                   session_duration=7, simulate_missing=False, random_seed=random_seed + i)
        t0 = time.time()
        res = fit(occu, **data, num_samples=num_samples, num_warmup=num_warmup, num_chains=1, timeout=3600)
        wall = time.time() - t0
        r = res.mcmc.result
        psi = res.samples["psi"][:, 0, :, 0].reshape(1, num_samples, n_sites)
        rows.append(dict(
            n_sites=n_sites, visits=time_periods, fit_wall_s=wall, kernel_ms=r.kernel_ms,
            leapfrogs=int(r.n_leapfrog.sum()), us_per_leapfrog=1e3 * r.kernel_ms / max(int(r.n_leapfrog.sum()), 1),
            wgs_per_chain=r.wgs_per_chain, lds_staged=bool(r.lds_staged),
            ess_psi_mean=float(effective_sample_size(psi).mean()), psi_mean=float(psi.mean()), true_occupancy=float(truth["z"].mean()),
        ))
    print(json.dumps(dict(
        methodology="biolith/benchmarks/occu_spoccupancy.py:16-118,436-452 (biolith leg): fit(occu) wall time, 1 chain, 500 draws + 100 warmup",
        note="row 0 includes loading libbiolith_hip.so", rows=rows), indent=1))


if __name__ == "__main__":
    main()
