#!/usr/bin/env python3
"""The top rows of the reference's benchmark grid (benchmarks/occu_spoccupancy.py:16-70: 6 400 x 64 and 12 800 x 90, 2 + 1 covariates,
one chain, 100 warmup + 500 draws) over workgroup counts and lane groups of the WIDE geometry (a chain across XCDs, fabric exchange):
   python tools/time_wide.py [row ...]          rows 6 and 7 by default
Forces (k, G) through BIOLITH_HIP_WIDE_K / BIOLITH_HIP_OCCU_G (A/B knobs of choose_geometry); the first line of a row is the host's own choice."""
import contextlib
import io
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from biolith_amd.engine import OccuDataset  # noqa: E402
from biolith_amd.models import simulate  # noqa: E402


def run(ds, label):
    best = None
    for s in range(2):
        r = ds.nuts(num_warmup=100, num_samples=500, num_chains=1, seed=s)
        us = 1e3 * r.kernel_ms / max(int(r.n_leapfrog.sum()) + 1, 1)
        best = us if best is None else min(best, us)
    print(f"  {label:28s} k={r.wgs_per_chain:4d} lanes/pair={r.lane_group} lds={r.lds_bytes:7d} staged={r.lds_staged} {best:7.3f} us/leapfrog  kernel {r.kernel_ms:.1f} ms  div {int(r.diverging.sum())}", flush=True)
    return best


def main():
    rows = [int(a) for a in sys.argv[1:]] or [6, 7]
    for i in rows:
        n_sites, visits = int(100 * 2 ** i), int(8 * 2 ** (i / 2))
        with contextlib.redirect_stdout(io.StringIO()):
            data, _ = simulate(n_site_covs=2, n_obs_covs=1, n_sites=n_sites, deployment_days_per_site=visits * 7, session_duration=7,
                               simulate_missing=False, random_seed=42 + i)
        ds = OccuDataset(data["site_covs"], data["obs_covs"], data["obs"])
        print(f"row {i}: {n_sites} x {visits}", flush=True)
        for v in ("BIOLITH_HIP_WIDE_K", "BIOLITH_HIP_OCCU_G"):
            os.environ.pop(v, None)
        run(ds, "host's choice")
        for k in ([48, 64, 96, 128, 192, 256] if i == 6 else [66, 96, 128, 192, 256]):
            for G in (1, 2, 4, 8, 16):
                if (n_sites + 1) // 2 * G > k * 256:
                    continue
                os.environ["BIOLITH_HIP_WIDE_K"], os.environ["BIOLITH_HIP_OCCU_G"] = str(k), str(G)
                try:
                    run(ds, f"forced k={k} G={G}")
                except Exception as exc:  # noqa: BLE001
                    print(f"  forced k={k} G={G}: {type(exc).__name__}: {exc}", flush=True)
        ds.close()


if __name__ == "__main__":
    main()
