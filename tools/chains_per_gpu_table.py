#!/usr/bin/env python3
"""Fold gpurun_out/chains_per_gpu/c<C>.json (tools/chains_per_gpu.sh: bench.py --chains-per-gpu C --no-secondary) into
profiles/r05/chains_per_gpu.json -- the chains-per-GPU capacity curve of ONE MI355X."""
import json
import os
import sys

src = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/chains_per_gpu"
dst = sys.argv[2] if len(sys.argv) > 2 else "profiles/r05/chains_per_gpu.json"
rows = []
for C in (1, 2, 4, 8, 16, 32):
    f = os.path.join(src, f"c{C}.json")
    if not os.path.exists(f):
        continue
    line = [ln for ln in open(f) if ln.startswith("{")][-1]
    o = json.loads(line)
    r = o["roofline"]
    rows.append(dict(chains_per_gpu=C, ess_per_s=o["value"], ms_per_step=o["ms_per_step"], kernel_ms=o["kernel_ms_per_rank"][0],
                     us_per_leapfrog_per_chain=r["us_per_leapfrog_per_chain"], us_per_leapfrog_slowest_chain=r.get("us_per_leapfrog_slowest_chain"),
                     wgs_per_chain=o["config"]["wgs_per_chain"], workgroups=C * o["config"]["wgs_per_chain"], kernel=r["kernel"],
                     effective_GBps=r["achieved"], max_split_rhat=o["ess"]["max_split_rhat"]))
json.dump(dict(workload="occu 10 000 x 5, 3 + 3 covariates, C chains x (1000 + 1000), 5 timed steps", command="bash tools/chains_per_gpu.sh",
               rows=rows), open(dst, "w"), indent=1)
for r in rows:
    print(r["chains_per_gpu"], round(r["ess_per_s"]), round(r["ms_per_step"], 2), round(r["us_per_leapfrog_per_chain"], 3), r["wgs_per_chain"], r["kernel"])
