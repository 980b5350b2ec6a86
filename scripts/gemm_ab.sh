#!/bin/bash
out=gpurun_out/${1:-gemm_ab}.txt; : > $out
step() { python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-profile "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'])"; }
for rep in 1 2; do
  for v in 0 1; do
    echo "== MI355_GEMM256=$v ResNetUnet bf16 bs32" >> $out; MI355_GEMM256=$v step --model ResNetUnet --batch 32 --dtype bf16 >> $out
    echo "== MI355_GEMM256=$v AttentionUNet bf16 bs32" >> $out; MI355_GEMM256=$v step >> $out
    echo "== MI355_GEMM256=$v R2AttU_Net bf16 bs16" >> $out; MI355_GEMM256=$v step --model R2AttU_Net --batch 16 >> $out
  done
done
paste - - < $out
