#!/usr/bin/env python3
"""Summarise the rocprofv3 --pmc passes of tools/pmc_run.sh: per-launch means of every counter for
the NUTS kernel, and the HBM byte figure bench.py reports as roofline.traffic.
   python tools/pmc_summary.py gpurun_out/pmc profiles/r01/d_pmc_summary.json [kernel name substring, default bl_nuts_kernel]
HBM bytes per launch = 2 x FETCH_SIZE KB (gfx950 correction of /opt/skills/guides/MI355X_MICROARCH.md,
an upper bound for this kernel's narrow staging loads) + WRITE_SIZE KB."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict


def main():
    src, dst = sys.argv[1], sys.argv[2]
    which = sys.argv[3] if len(sys.argv) > 3 else "bl_nuts_kernel"
    sums, counts = defaultdict(float), defaultdict(int)
    kernel = None
    for path in glob.glob(os.path.join(src, "**", "*counter_collection.csv"), recursive=True):
        per_dispatch = defaultdict(float)
        with open(path) as f:
            for row in csv.DictReader(f):
                if which not in row["Kernel_Name"]:
                    continue
                kernel = row["Kernel_Name"]
                per_dispatch[(row["Dispatch_Id"], row["Counter_Name"])] += float(row["Counter_Value"])
        for (_, name), v in per_dispatch.items():
            sums[name] += v
            counts[name] += 1
    mean = {k: sums[k] / counts[k] for k in sorted(sums)}
    fetch_kb, write_kb = mean.get("FETCH_SIZE", 0.0), mean.get("WRITE_SIZE", 0.0)
    out = dict(
        kernel=kernel,
        command="python bench.py --steps 2 --warmup 1 --no-cpu-baseline (tools/pmc_run.sh: one rocprofv3 --pmc pass per counter set)",
        per_launch_mean=mean, FETCH_SIZE_KB=fetch_kb, WRITE_SIZE_KB=write_kb,
        hbm_bytes_per_launch=(2.0 * fetch_kb + write_kb) * 1024.0,
        l2_hit_rate=(mean["TCC_HIT_sum"] / (mean["TCC_HIT_sum"] + mean["TCC_MISS_sum"])) if "TCC_HIT_sum" in mean else None,
        note="FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 reports half the bytes of a wide coalesced read; this kernel's "
             "staging loads are 4 B/lane, so the doubled figure is an upper bound).  The dataset is read from HBM once and then "
             "lives in LDS; everything else is exchange granules that hit in L2.",
    )
    with open(dst, "w") as f:
        json.dump(out, f, indent=1)
    print(json.dumps({k: out[k] for k in ("kernel", "FETCH_SIZE_KB", "WRITE_SIZE_KB", "hbm_bytes_per_launch", "l2_hit_rate")}))


if __name__ == "__main__":
    main()
