#!/bin/bash
# Chains-per-GPU capacity curve on ONE GPU: python bench.py --chains-per-gpu C --no-secondary for C in 1..32.
# Writes gpurun_out/chains_per_gpu/c<C>.json (the bench line) ; tools/chains_per_gpu_table.py folds them into profiles/r05/chains_per_gpu.json
out=gpurun_out/chains_per_gpu; mkdir -p $out
for C in 1 2 4 8 16 32; do
  timeout 600 python bench.py --chains-per-gpu $C --no-secondary --no-cpu-baseline --no-e2e --no-live-pmc --full-line --steps 5 --warmup 1 > $out/c$C.json 2> $out/c$C.err || echo "C=$C failed rc=$?" >> $out/failures.txt
done
