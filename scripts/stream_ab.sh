#!/bin/bash
# Streaming-pass A/B on one box: reversed sweep of the BatchNorm backward apply pass, rows in flight in the reduce pass, workgroups
# per CU of the row reductions.  Step time (three interleaved rounds) + the kernel-table rows of the passes per variant.
R=$PWD
BENCH_ARGS="" bash scripts/env_sweep.sh r04h_stream_ab "" "MI355_BN_APPLY_REV=1" "MI355_LIB=$R/ab/redf8.so" "MI355_LIB=$R/ab/redf2.so" "MI355_RR_WGS=512" "MI355_BN_APPLY_REV=1 MI355_LIB=$R/ab/redf8.so"
for v in "" "MI355_BN_APPLY_REV=1" "MI355_LIB=$R/ab/redf8.so" "MI355_RR_WGS=512"; do
  echo "== [${v:-defaults}]" >> gpurun_out/r04h_stream_ab.txt
  env $v python bench.py --steps 10 --warmup 3 --no-cpu-baseline --kernel-table 2>&1 >/dev/null | grep -E "bn_bwd|gate_bn|bn_act|sum of plan" >> gpurun_out/r04h_stream_ab.txt
done
cat gpurun_out/r04h_stream_ab.txt
