#!/bin/bash
# timing-only builds of the weight-gradient kernels (stale rows / no barrier): where the row step's time goes
set -o pipefail
out=gpurun_out/${1:-wgrad_probe}.txt
: > $out
L="32,256,256,64,64,0,1 32,128,128,128,128,0,1 32,64,64,512,256,0,1 32,32,32,512,512,0,1 32,32,32,1024,512,0,1"
for lib in "" ab/w8_nodma.so ab/w8_nobar.so ab/w8_none.so; do
  for v in 1 0; do
    echo "== lib=${lib:-default} MI355_WGRAD8=$v" >> $out
    MI355_LIB=${lib:+$PWD/$lib} MI355_WGRAD8=$v python scripts/conv_bench.py $L 2>/dev/null >> $out || exit 1
  done
done
echo "== default lib, four-wave kernel at two workgroups per CU (MI355_WGRAD_WGS=512)" >> $out
MI355_WGRAD8=0 MI355_WGRAD_WGS=512 python scripts/conv_bench.py $L 2>/dev/null >> $out || exit 1
cat $out
