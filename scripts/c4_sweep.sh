#!/bin/bash
# R2AttU_Net 256x256 bs 16 bf16 (config C4): step time against the eight-wave weight gradient's grid and the 128-channel kernel's grid-fill rule
out=gpurun_out/${1:-c4_sweep}.txt
: > $out
step() { python bench.py --model R2AttU_Net --batch 16 --steps 8 --warmup 3 --no-cpu-baseline --no-profile 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'])"; }
for rep in 1 2; do
  echo "== four-wave wgrad (256)" >> $out; MI355_WGRAD8=0 step >> $out
  for w in 96 128 192 256; do echo "== eight-wave $w" >> $out; MI355_WGRAD_WGS=$w step >> $out; done
  echo "== eight-wave 128, pp128 fill 50" >> $out; MI355_PP128_FILL=50 step >> $out
  echo "== eight-wave 256, pp128 fill 50" >> $out; MI355_WGRAD_WGS=256 MI355_PP128_FILL=50 step >> $out
done
cat $out
