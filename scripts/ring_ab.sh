#!/bin/bash
# ring prefetch in the row-reduction skeleton (HEAD) against batch-then-drain (ab/noring.so): step and streaming rows, C3 and C5's classifier
R=$PWD
bash scripts/env_sweep.sh r04k_ring2 "" "MI355_LIB=$R/ab/noring.so"
BENCH_ARGS="--model vgg16_bn --batch 16 --size 512 --dtype fp16" bash scripts/env_sweep.sh r04k_ring2_c5 "" "MI355_LIB=$R/ab/noring.so"
for v in "" "MI355_LIB=$R/ab/noring.so"; do
  echo "== [${v:-defaults}]" >> gpurun_out/r04k_ring2.txt
  env $v python bench.py --steps 10 --warmup 3 --no-cpu-baseline --kernel-table 2>&1 >/dev/null | grep -E "pool2|sum of plan" >> gpurun_out/r04k_ring2.txt
  echo "== C5 [${v:-defaults}]" >> gpurun_out/r04k_ring2.txt
  env $v python bench.py --model vgg16_bn --batch 16 --size 512 --dtype fp16 --steps 10 --warmup 3 --no-cpu-baseline --kernel-table 2>&1 >/dev/null | grep -E "pool2|bn_bwd|sum of plan" >> gpurun_out/r04k_ring2.txt
done
cat gpurun_out/r04k_ring2.txt gpurun_out/r04k_ring2_c5.txt
