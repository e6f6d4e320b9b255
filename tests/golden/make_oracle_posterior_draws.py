#!/usr/bin/env python3
"""Draws of the CPU oracle's NUTS for the posterior-parity tests whose oracle leg is the slow one on the GPU box (VERDICT r05 weak 10:
`test_nmix_posterior_matches_oracle` 42 s, `test_nmix_re_posterior_matches_oracle` 61 s -- oracle CPU time, not GPU) -> small .npz
fixtures (float32 draws of the compared coordinates; inputs are the committed golden data sets, so the fixture is data, not code).

    python tests/golden/make_oracle_posterior_draws.py        (a minute or two on 4 cores)

Each entry: the golden data set's name, the oracle's keyword arguments, warmup / draws / seed, and how many leading coordinates are kept."""
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
sys.path.insert(0, os.path.dirname(HERE))
import oracle  # noqa: E402
from conftest import PARITY_S, PARITY_W, load_golden  # noqa: E402

CASES = {
    # tests/test_gpu_nmix.py::test_nmix_posterior_matches_oracle
    "nmix_small_2x2": dict(data="nmix_small_2x2", kw=dict(model="nmixture", max_abundance=40), warmup=PARITY_W, samples=PARITY_S, seed=0, keep=None),
    # tests/test_gpu_nmix_re.py::test_nmix_re_posterior_matches_oracle: the fixed effects and log site_re_sd (K = the largest count + 5)
    "nmix_site_re": dict(data="nmix_site_re", kw=dict(model="nmixture", max_abundance=None, site_random_effects=True, obs_random_effects=False,
                                                       prior_site_re_sd=0.8, prior_obs_re_sd=1.2), warmup=500, samples=1000, seed=0, keep="fixed+1"),
}


def main():
    for name, c in CASES.items():
        g = load_golden(c["data"])
        kw = dict(c["kw"])
        if kw.get("max_abundance") is None:
            kw["max_abundance"] = int(np.nanmax(g["obs"])) + 5
        od = oracle.OracleData(g["site_covs"], g["obs_covs"], g["obs"], **kw)
        t0 = time.time()
        o = oracle.nuts_run(od, c["warmup"], c["samples"], num_chains=4, seed=c["seed"])
        keep = od.D if c["keep"] is None else od.Ks + od.Ko + 3
        path = os.path.join(HERE, f"oracle_draws_{name}.npz")
        np.savez_compressed(path, draws=o["draws"][:, :, :keep].astype(np.float32), D=od.D, warmup=c["warmup"], samples=c["samples"], seed=c["seed"],
                            max_abundance=kw["max_abundance"], divergences=int(o["diverging"].sum()))
        print(f"{name}: D {od.D}, kept {keep}, {time.time() - t0:.0f} s, {os.path.getsize(path) / 1024:.0f} KB")


if __name__ == "__main__":
    main()
