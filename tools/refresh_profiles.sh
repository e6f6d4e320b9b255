#!/bin/bash
# Round-end artefacts on the GPU box (run from the repo root through gpurun): bench lines, rocprofv3 kernel stats and PMC passes of the
# bench workloads on the library as built.   usage: [LIGHT=1] bash tools/refresh_profiles.sh <outdir under gpurun_out/>
# LIGHT=1: bench lines, kernel stats and PMC passes only (no timing tools, stamps, grid or CPU-baseline validation run)
set -u
LIGHT=${LIGHT:-0}
OUT=${1:-gpurun_out/refresh}
mkdir -p "$OUT"
ROOT=$(pwd)
python bench.py --full-out "$OUT/bench_full.json" > "$OUT/bench.json" 2> "$OUT/bench.err"; echo "bench (default: headline + secondary) rc=$?"
for wl in occu_rn occu_re occu_stacked occu_dyn occu_cfg1; do
  python bench.py --workload $wl --steps 3 --no-e2e --full-out "$OUT/bench_full_$wl.json" > "$OUT/bench_$wl.json" 2> "$OUT/bench_$wl.err"; echo "bench $wl rc=$?"
done
if [ "$LIGHT" != 1 ]; then
python tools/time_models.py > "$OUT/time_models.txt" 2>&1
python tools/time_re.py > "$OUT/time_re.txt" 2>&1
python tools/time_fit_e2e.py > "$OUT/time_fit_e2e.txt" 2>&1
python tools/time_rn.py > "$OUT/time_rn.txt" 2>&1
python tools/time_dyn.py > "$OUT/time_dyn.txt" 2>&1
python tools/time_occu_g.py > "$OUT/time_occu_g.txt" 2>&1
python benchmarks/fit_time_grid.py > "$OUT/fit_time_grid.json" 2> "$OUT/fit_time_grid.err"
# in-kernel stamps (diagnostic builds: make -C biolith_amd/csrc stamps), when that library travelled along
if [ -f biolith_amd/lib/libbiolith_hip_stamps.so ]; then
  python tools/stamps.py > "$OUT/stamps.txt" 2>&1
  python tools/stamps_re.py > "$OUT/stamps_re.txt" 2>&1
  for m in occu_stacked occu_dyn; do python tools/stamps_model.py $m; done > "$OUT/stamps_models.txt" 2>&1
fi
# the scaled CPU baselines validated once: the oracle's own (short) sampler run beside the scaled estimate (minutes)
python bench.py --workload occu_rn --steps 2 --no-e2e --cpu-baseline --full-line > "$OUT/cpu_baseline_validation_rn.json" 2> "$OUT/cpu_baseline_validation_rn.err"; echo "cpu baseline validation rc=$?"
fi
cd /tmp && export TMPDIR=/tmp
for wl in occu occu_rn occu_re occu_stacked occu_dyn occu_cfg1; do
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$ROOT/$OUT/stats_$wl" -- python3 "$ROOT/bench.py" --workload $wl --steps 3 --no-cpu-baseline --no-e2e --no-secondary --no-live-pmc --full-line > "$ROOT/$OUT/bench_${wl}_under_rocprof.json" 2> "$ROOT/$OUT/stats_$wl.err"
  echo "stats $wl rc=$?"
done
cd "$ROOT"
bash tools/pmc_run.sh "$OUT/pmc_occu"
bash tools/pmc_run.sh "$OUT/pmc_rn" --workload occu_rn
bash tools/pmc_run.sh "$OUT/pmc_re" --workload occu_re
bash tools/pmc_run.sh "$OUT/pmc_dyn" --workload occu_dyn
bash tools/pmc_run.sh "$OUT/pmc_stacked" --workload occu_stacked
python tools/pmc_summary.py "$OUT/pmc_occu" "$OUT/pmc_summary_occu.json"
python tools/pmc_summary.py "$OUT/pmc_rn" "$OUT/pmc_summary_rn.json"
python tools/pmc_summary.py "$OUT/pmc_re" "$OUT/pmc_summary_re.json" bl_re_nuts_kernel
python tools/pmc_summary.py "$OUT/pmc_dyn" "$OUT/pmc_summary_dyn.json"
python tools/pmc_summary.py "$OUT/pmc_stacked" "$OUT/pmc_summary_stacked.json"
find "$OUT" -name "*kernel_stats.csv" | head
