#!/bin/bash
# step time against the eight-wave weight-gradient kernel's grid (workgroups, one per CU, each owning its CU)
set -o pipefail
out=gpurun_out/${1:-wgs_sweep8}.txt
: > $out
step() { python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-profile 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'])"; }
for rep in 1 2 3; do
  echo "== four-wave 256" >> $out; MI355_WGRAD8=0 step >> $out || exit 1
  for w in 64 96 112 128 144 160 192; do
    echo "== eight-wave $w" >> $out; MI355_WGRAD_WGS=$w step >> $out || exit 1
  done
done
cat $out
