#!/usr/bin/env python3
"""Copy what tools/refresh_profiles.sh left under gpurun_out/<dir> (scratch) into profiles/<round>/ (tracked) under one prefix:
    python tools/collect_profiles.py gpurun_out/r03 profiles/r03 a [workload ...]
bench lines, the rocprofv3 kernel-stats CSV of each workload's command, the PMC summaries, the timing scripts' outputs."""
import glob
import os
import shutil
import sys


def main():
    src, dst, pre = sys.argv[1], sys.argv[2], sys.argv[3]
    only = set(sys.argv[4:])
    os.makedirs(dst, exist_ok=True)
    short = {"occu": "", "occu_rn": "_rn", "occu_re": "_re", "occu_stacked": "_stacked", "occu_dyn": "_dyn", "occu_cfg1": "_cfg1"}

    def cp(a, b):
        if os.path.exists(a):
            shutil.copyfile(a, os.path.join(dst, f"{pre}_{b}"))
            print(f"{pre}_{b}")

    if not only or "occu" in only:
        cp(os.path.join(src, "bench.json"), "bench.json")            # the compact line the driver reads
        cp(os.path.join(src, "bench_full.json"), "bench_full.json")  # the verbose record beside it
    for wl, sfx in short.items():
        if only and wl not in only:
            continue
        if wl != "occu":
            cp(os.path.join(src, f"bench_{wl}.json"), f"bench{sfx}.json")
            cp(os.path.join(src, f"bench_full_{wl}.json"), f"bench_full{sfx}.json")
        cp(os.path.join(src, f"bench_{wl}_under_rocprof.json"), f"bench{sfx}_under_rocprof.json")
        # (gpurun MERGES into an existing directory: the newest file is this run's)
        for f in sorted(glob.glob(os.path.join(src, f"stats_{wl}", "**", "*kernel_stats.csv"), recursive=True), key=os.path.getmtime)[-1:]:
            cp(f, f"kernel_stats{sfx}.csv")
        key = {"occu": "occu", "occu_rn": "rn", "occu_re": "re", "occu_dyn": "dyn", "occu_stacked": "stacked"}.get(wl)
        if key:
            cp(os.path.join(src, f"pmc_summary_{key}.json"), f"pmc_summary{sfx}.json")
    if not only:
        for name in ("time_models.txt", "time_re.txt", "time_fit_e2e.txt", "time_rn.txt", "stamps_rn.txt", "time_dyn.txt", "time_occu_g.txt",
                     "fit_time_grid.json", "stamps.txt", "stamps_re.txt", "stamps_models.txt", "cpu_baseline_validation_rn.json"):
            cp(os.path.join(src, name), name)


if __name__ == "__main__":
    main()
