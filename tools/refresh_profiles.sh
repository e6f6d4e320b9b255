#!/bin/bash
# Round-end artefacts on the GPU box (run from the repo root through gpurun): bench lines, rocprofv3 kernel stats and PMC passes of the
# three bench workloads on the library as built.   usage: bash tools/refresh_profiles.sh <outdir under gpurun_out/>
set -u
OUT=${1:-gpurun_out/refresh}
mkdir -p "$OUT"
ROOT=$(pwd)
python bench.py > "$OUT/bench.json" 2> "$OUT/bench.err"; echo "bench rc=$?"
python bench.py --workload occu_re --no-e2e > "$OUT/bench_re.json" 2> "$OUT/bench_re.err"; echo "bench_re rc=$?"
python bench.py --workload occu_rn --no-e2e > "$OUT/bench_rn.json" 2> "$OUT/bench_rn.err"; echo "bench_rn rc=$?"
python tools/time_models.py > "$OUT/time_models.txt" 2>&1
python tools/time_re.py > "$OUT/time_re.txt" 2>&1
python tools/time_fit_e2e.py > "$OUT/time_fit_e2e.txt" 2>&1
cd /tmp && export TMPDIR=/tmp
for wl in occu occu_re; do
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$ROOT/$OUT/stats_$wl" -- python3 "$ROOT/bench.py" --workload $wl --no-cpu-baseline --no-e2e > "$ROOT/$OUT/bench_${wl}_under_rocprof.json" 2> "$ROOT/$OUT/stats_$wl.err"
  echo "stats $wl rc=$?"
done
cd "$ROOT"
bash tools/pmc_run.sh "$OUT/pmc_occu"
bash tools/pmc_run.sh "$OUT/pmc_re" --workload occu_re
python tools/pmc_summary.py "$OUT/pmc_occu" "$OUT/pmc_summary_occu.json"
python tools/pmc_summary.py "$OUT/pmc_re" "$OUT/pmc_summary_re.json" bl_re_nuts_kernel
find "$OUT" -name "*kernel_stats.csv" | head
