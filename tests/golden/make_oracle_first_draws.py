"""Regenerates oracle_first_draws.json (see its "note").  Run from the repo root: python tests/golden/make_oracle_first_draws.py"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle  # noqa: E402
from conftest import load_golden  # noqa: E402

g = load_golden("small_3x3")
out = {}
od = oracle.OracleData(g["site_covs"], g["obs_covs"], g["obs"])
r = oracle.nuts_run(od, 20, 3, num_chains=3, seed=3)
out["occu"] = dict(draws=r["draws"].tolist(), num_steps=r["num_steps"].tolist(), step_size=np.asarray(r["step_size"]).tolist())
od = oracle.OracleData(g["site_covs"], g["obs_covs"], g["obs"], model="occu_re", site_random_effects=True)
r = oracle.nuts_run(od, 10, 2, num_chains=2, seed=3)
out["occu_re_site"] = dict(draws=r["draws"][:, :, :12].tolist(), num_steps=r["num_steps"].tolist(), step_size=np.asarray(r["step_size"]).tolist())
note = ("oracle.nuts_run on simulate_small_3x3: occu 20 warmup + 3 draws, 3 chains, seed 3; occu_re (site effects) 10 + 2, 2 chains, "
        "first 12 coordinates. Written by tests/golden/make_oracle_first_draws.py from the oracle whose trees the GPU kernels reproduce; "
        "pins the RNG stream layout (64 streams per chain, scalar 63, direction 62; D + 2 with random effects).")
json.dump(dict(note=note, **out), open(os.path.join(ROOT, "tests", "golden", "oracle_first_draws.json"), "w"), indent=1)
