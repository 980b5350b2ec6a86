#!/bin/bash
set -o pipefail
out=gpurun_out/${1:-wgrad_probe2}.txt
: > $out
MI355_LIB=$PWD/ab/w8_stagger.so python -m pytest tests/test_gpu_conv.py -x -q -m gpu -k "test_conv_wgrad" 2>&1 | tail -3 >> $out || { cat $out; exit 1; }
L="32,256,256,64,64,0,1 32,256,256,128,64,0,1 32,128,128,128,128,0,1 32,64,64,512,256,0,1 32,32,32,512,512,0,1 32,32,32,1024,512,0,1"
for lib in "" ab/w8_stagger.so "" ab/w8_stagger.so; do
  echo "== lib=${lib:-default} MI355_WGRAD8=1" >> $out
  MI355_LIB=${lib:+$PWD/$lib} MI355_WGRAD8=1 python scripts/conv_bench.py $L 2>/dev/null | awk '{print $1, $7, "ms", $(NF-5), "TFLOP/s"}' >> $out || exit 1
done
step() { python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-profile 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'])"; }
for rep in 1 2; do
  echo "== step: four-wave" >> $out; MI355_WGRAD8=0 step >> $out || exit 1
  echo "== step: eight-wave stagger, 256 wgs" >> $out; MI355_LIB=$PWD/ab/w8_stagger.so step >> $out || exit 1
  echo "== step: eight-wave stagger, 192 wgs" >> $out; MI355_WGRAD_WGS=192 MI355_LIB=$PWD/ab/w8_stagger.so step >> $out || exit 1
  echo "== step: eight-wave stagger, 128 wgs" >> $out; MI355_WGRAD_WGS=128 MI355_LIB=$PWD/ab/w8_stagger.so step >> $out || exit 1
  echo "== step: eight-wave, serial streams" >> $out; python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-profile --serial-streams 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'])" >> $out || exit 1
  echo "== step: four-wave, serial streams" >> $out; MI355_WGRAD8=0 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-profile --serial-streams 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'])" >> $out || exit 1
done
cat $out
