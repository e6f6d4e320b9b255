set -u
OUT=gpurun_out/r05o; mkdir -p $OUT; ROOT=$(pwd)
python bench.py --workload occu_rn --steps 3 --no-e2e --full-line > $OUT/bench_occu_rn.json 2> $OUT/bench_occu_rn.err
python tools/time_rn.py > $OUT/time_rn.txt 2>&1
python tools/stamps_rn.py > $OUT/stamps_rn.txt 2>&1
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$ROOT/$OUT/stats_occu_rn" -- python3 "$ROOT/bench.py" --workload occu_rn --steps 3 --no-cpu-baseline --no-e2e --no-secondary --no-live-pmc --full-line > "$ROOT/$OUT/bench_occu_rn_under_rocprof.json" 2> "$ROOT/$OUT/stats_occu_rn.err"
cd "$ROOT"
bash tools/pmc_run.sh "$OUT/pmc_rn" --workload occu_rn > /dev/null 2>&1
python tools/pmc_summary.py "$OUT/pmc_rn" "$OUT/pmc_summary_rn.json" > /dev/null 2>&1
python bench.py > $OUT/bench.json 2> $OUT/bench.err
tail -c 400 $OUT/bench.json
