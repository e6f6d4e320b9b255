#!/bin/bash
# PMC passes for the NUTS kernel (separate passes: TCC FETCH_SIZE and WRITE_SIZE do not fit one pass).
# usage (on the GPU box): bash tools/pmc_run.sh <outdir> [extra bench.py arguments, e.g. --workload occu_re]
set -u
OUT=${1:-gpurun_out/pmc}
shift || true
EXTRA="$*"
mkdir -p "$OUT"
ROOT=$(pwd)
cd /tmp && export TMPDIR=/tmp
for set in "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_ANY" "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum"; do
  tag=$(echo $set | tr ' ' '_' | cut -c1-40)
  timeout 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d "$ROOT/$OUT/$tag" -- python3 "$ROOT/bench.py" --steps 2 --warmup 1 --no-cpu-baseline --no-e2e --no-secondary --no-live-pmc --full-line $EXTRA > "$ROOT/$OUT/$tag.json" 2> "$ROOT/$OUT/$tag.err"
  echo "$tag rc=$?"
done
