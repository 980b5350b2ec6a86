#!/bin/bash
# step time + the streaming rows of the kernel table, C3 and C4 (tag = $1)
t=$1
bash scripts/env_sweep.sh ${t}_c3 ""
BENCH_ARGS="--model R2AttU_Net --batch 16" bash scripts/env_sweep.sh ${t}_c4 ""
python bench.py --steps 10 --warmup 3 --no-cpu-baseline --kernel-table 2>&1 >/dev/null | grep -E "bn_|gate_|sum of plan" > gpurun_out/${t}_tbl.txt
echo "== C4" >> gpurun_out/${t}_tbl.txt
python bench.py --model R2AttU_Net --batch 16 --steps 10 --warmup 3 --no-cpu-baseline --kernel-table 2>&1 >/dev/null | grep -E "bn_|gate_|sum of plan" >> gpurun_out/${t}_tbl.txt
cat gpurun_out/${t}_tbl.txt
