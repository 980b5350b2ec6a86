#!/bin/bash
R=$PWD
bash scripts/env_sweep.sh r04i_redw "" "MI355_LIB=$R/ab/redw512.so" "MI355_LIB=$R/ab/redw1024.so"
for v in "" "MI355_LIB=$R/ab/redw512.so" "MI355_LIB=$R/ab/redw1024.so"; do
  echo "== [${v:-defaults}]" >> gpurun_out/r04i_redw.txt
  env $v python bench.py --steps 10 --warmup 3 --no-cpu-baseline --kernel-table 2>&1 >/dev/null | grep -E "bn_bwd_reduce|sum of plan" >> gpurun_out/r04i_redw.txt
done
cat gpurun_out/r04i_redw.txt
