#!/usr/bin/env python3
"""Phase breakdown of one leapfrog tick from the BL_STAMPS diagnostic build (shares, not run time).
   make -C biolith_amd/csrc stamps && BIOLITH_HIP_LIB=biolith_amd/lib/libbiolith_hip_stamps.so python tools/stamps.py"""
import contextlib
import io
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("BIOLITH_HIP_LIB", os.path.join(ROOT, "biolith_amd", "lib", "libbiolith_hip_stamps.so"))
from biolith_amd.engine import OccuDataset  # noqa: E402
from biolith_amd.models import simulate  # noqa: E402

NAMES = ["decisions + bookkeeping (under phase A)", "wait for compute waves", "wg partial+publish", "sweep (poll)",
         "speculative position", "barrier2", "-", "-"]


def main():
    with contextlib.redirect_stdout(io.StringIO()):
        data, _ = simulate(n_sites=10000, n_site_covs=3, n_obs_covs=3, deployment_days_per_site=35, session_duration=7)
    ds = OccuDataset(data["site_covs"], data["obs_covs"], data["obs"])
    ks = [int(a) for a in sys.argv[1:]] or [0]
    for k in ks:
        for chains in (4,):
            r = ds.nuts(num_warmup=1000, num_samples=1000, num_chains=chains, seed=0, wgs_per_chain=k)
            c = ds.debug_counters()
            ticks, rt = int(c[8]), int(c[9])
            tot = c[:7].sum()
            mhz = tot / (rt / 100.0) if rt else float("nan")  # s_memrealtime ticks at 100 MHz
            print(f"k={r.wgs_per_chain} chains={chains} kernel {r.kernel_ms:.1f} ms ticks {ticks} "
                  f"cycles/tick {tot / max(ticks, 1):.0f} clock {mhz:.0f} MHz us/tick {r.kernel_ms * 1e3 / max(ticks, 1):.2f} l2local_chains {r.chains_l2_local} poll-rounds/tick {int(c[10]) / max(ticks, 1) + 1:.2f}")
            for n, v in zip(NAMES[:6], c[:6]):
                print(f"    {n:40s} {v / max(ticks, 1):8.0f} cyc  {100.0 * v / tot:5.1f} %")
            kn = c[11:14].astype(float)
            kc = np.array([c[14], c[15], c[7]], dtype=float)
            print(f"    site evaluation (compute wave 1): {int(c[20])} passes, {c[21] / max(float(c[20]), 1.0):.0f} cyc each")
            rounds = float(c[10]) + ticks
            print(f"    poll rounds: {rounds / max(ticks, 1):.2f} per exchange, {c[19] / max(rounds, 1.0):.0f} cyc each (loads issued -> tags checked)")
            n_end = max(float(c[13]), 1.0)
            print("    transition end, per tick: flush bookkeeping %.0f | select/adapt/output %.0f | new momentum+tree %.0f cyc"
                  % (c[16] / n_end, c[17] / n_end, c[18] / n_end))
            for name, n_, cyc in zip(("next leaf of the subtree", "next doubling", "transition end / init"), kn, kc):
                print(f"    decisions, {name:26s}: {int(n_):7d} ticks ({100 * n_ / max(ticks, 1):4.1f} %), {cyc / max(n_, 1):7.0f} cyc each")
            # the same counters of a launch that is warm-up only (1000 + 2), subtracted -> the sampling phase alone:
            # the adaptation's share of a transition's end (dual averaging, Welford moments, window ends) must be gone there
            rw = ds.nuts(num_warmup=1000, num_samples=2, num_chains=chains, seed=0, wgs_per_chain=k)
            cw = ds.debug_counters()
            n_end_w, n_end_s = max(float(cw[13]), 1.0), max(float(c[13] - cw[13]), 1.0)
            for label, cc, ne in (("warm-up (1000 transitions)", cw, n_end_w), ("sampling (the other 998)", c - cw, n_end_s)):
                tk = max(float(cc[8]), 1.0)
                print(f"    {label:28s}: {int(cc[8]):6d} ticks, {cc[:7].sum() / tk:6.0f} cyc per tick; transition end {float(cc[7]) / ne:6.0f} cyc = flush bookkeeping {cc[16] / ne:.0f} | "
                      f"select/adapt/output {cc[17] / ne:.0f} | new momentum+tree {cc[18] / ne:.0f} | the leaf's own decisions {(float(cc[7]) - cc[16] - cc[17] - cc[18]) / ne:.0f}")


if __name__ == "__main__":
    main()
