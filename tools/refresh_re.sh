set -u
OUT=gpurun_out/f; mkdir -p $OUT; ROOT=$(pwd)
python bench.py --workload occu_re --no-e2e --full-line > $OUT/bench_re.json 2> $OUT/bench_re.err; echo rc=$?
python tools/time_re.py > $OUT/time_re.txt 2>&1
python tools/time_models.py > $OUT/time_models.txt 2>&1
python tools/stamps_re.py > $OUT/stamps_re.txt 2>&1
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$ROOT/$OUT/stats_occu_re" -- python3 "$ROOT/bench.py" --workload occu_re --no-cpu-baseline --no-e2e --full-line > "$ROOT/$OUT/bench_occu_re_under_rocprof.json" 2> "$ROOT/$OUT/stats.err"; echo stats rc=$?
cd "$ROOT"
bash tools/pmc_run.sh "$OUT/pmc_re" --workload occu_re
python tools/pmc_summary.py "$OUT/pmc_re" "$OUT/pmc_summary_re.json" bl_re_nuts_kernel
