#!/bin/bash
# Same-box A/B of the weight-gradient kernels: per-layer timings (scripts/conv_bench.py) and the whole step, four-wave (MI355_WGRAD8=0) vs eight-wave.
set -o pipefail
out=gpurun_out/${1:-wgrad_ab}.txt
: > $out
L="32,256,256,64,64,0,1 32,256,256,128,64,0,1 32,128,128,128,64,1,1 32,128,128,64,128,0,1 32,128,128,128,128,0,1 32,128,128,256,128,0,1 32,64,64,256,256,0,1 32,64,64,512,256,0,1 32,32,32,512,512,0,1 32,32,32,1024,512,0,1 32,16,16,1024,512,1,1 16,256,256,64,64,0,1 16,32,32,512,512,0,1"
for v in 0 1 0 1; do
  echo "== MI355_WGRAD8=$v per layer" >> $out
  MI355_WGRAD8=$v python scripts/conv_bench.py $L 2>/dev/null >> $out || exit 1
done
for v in 0 1 0 1; do
  echo "== MI355_WGRAD8=$v step" >> $out
  MI355_WGRAD8=$v python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-profile 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'])" >> $out || exit 1
done
cat $out
