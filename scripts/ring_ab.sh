#!/bin/bash
# ring prefetch in the row-reduction skeleton (HEAD) against batch-then-drain (ab/noring.so): parity subset, step and streaming rows, C3 and C4
R=$PWD
python -m pytest tests -m gpu -x -q -k "bn or batchnorm or BatchNorm or gate or pool or fusions or blocks or ops or rowdot" 2>&1 | tail -2 || exit 1
bash scripts/env_sweep.sh r04k_ring "" "MI355_LIB=$R/ab/noring.so"
BENCH_ARGS="--model R2AttU_Net --batch 16" bash scripts/env_sweep.sh r04k_ring_c4 "" "MI355_LIB=$R/ab/noring.so"
for v in "" "MI355_LIB=$R/ab/noring.so"; do
  echo "== [${v:-defaults}]" >> gpurun_out/r04k_ring.txt
  env $v python bench.py --steps 10 --warmup 3 --no-cpu-baseline --kernel-table 2>&1 >/dev/null | grep -E "bn_bwd|gate_bn|sum of plan" >> gpurun_out/r04k_ring.txt
done
cat gpurun_out/r04k_ring.txt
