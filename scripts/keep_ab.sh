#!/bin/bash
# Same-box A/B: the library at HEAD (cache policy / optional operands of the streaming BatchNorm passes as template parameters, the
# recurrent block's d x summed once) against ab/old_bn.so (the previous bn.hip / gate.hip: run-time switches around the loads).
R=$PWD
bash scripts/env_sweep.sh r04h_keep_ab "" "MI355_DEFER_POST=0 MI355_LIB=$R/ab/old_bn.so"
BENCH_ARGS="--model R2AttU_Net --batch 16" bash scripts/env_sweep.sh r04h_keep_ab_c4 "" "MI355_DEFER_POST=0" "MI355_DEFER_POST=0 MI355_LIB=$R/ab/old_bn.so"
for v in "" "MI355_DEFER_POST=0 MI355_LIB=$R/ab/old_bn.so"; do
  echo "== [${v:-defaults}]" >> gpurun_out/r04h_keep_ab.txt
  env $v python bench.py --steps 10 --warmup 3 --no-cpu-baseline --kernel-table 2>&1 >/dev/null | grep -E "bn_|gate_|sum of plan" >> gpurun_out/r04h_keep_ab.txt
  echo "== C4 [${v:-defaults}]" >> gpurun_out/r04h_keep_ab_c4.txt
  env $v python bench.py --model R2AttU_Net --batch 16 --steps 10 --warmup 3 --no-cpu-baseline --kernel-table 2>&1 >/dev/null | grep -E "bn_|gate_|sum of plan" >> gpurun_out/r04h_keep_ab_c4.txt
done
cat gpurun_out/r04h_keep_ab.txt gpurun_out/r04h_keep_ab_c4.txt
