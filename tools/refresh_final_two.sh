#!/bin/bash
# The headline's and occu_rn's bench lines, rocprofv3 kernel stats and PMC passes on the library as built (a light form of refresh_profiles.sh).
set -u
OUT=${1:-gpurun_out/final2}
mkdir -p "$OUT"; ROOT=$(pwd)
python bench.py --workload occu_rn --steps 3 --no-e2e --full-out "$OUT/bench_full_occu_rn.json" > "$OUT/bench_occu_rn.json" 2> "$OUT/bench_occu_rn.err"; echo "bench rn rc=$?"
cd /tmp && export TMPDIR=/tmp
for wl in occu occu_rn; do
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$ROOT/$OUT/stats_$wl" -- python3 "$ROOT/bench.py" --workload $wl --steps 3 --no-cpu-baseline --no-e2e --no-secondary --no-live-pmc --full-line > "$ROOT/$OUT/bench_${wl}_under_rocprof.json" 2> "$ROOT/$OUT/stats_$wl.err"; echo "stats $wl rc=$?"
done
cd "$ROOT"
bash tools/pmc_run.sh "$OUT/pmc_rn" --workload occu_rn
python tools/pmc_summary.py "$OUT/pmc_rn" "$OUT/pmc_summary_rn.json"
