#!/bin/bash
# Clock the chip holds under each MFMA kernel, from dispatches LONG enough for GRBM_GUI_ACTIVE / 8 / duration to be trusted (>= 3 ms):
# the layers of the step at batches of 512-4096 through scripts/conv_bench.py.   scripts/clock_long.sh TAG   ->  gpurun_out/TAG_clock_long.txt
set -e -o pipefail
tag=$1
out=$PWD/gpurun_out/prof_$tag
mkdir -p $out
export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES --output-format csv -d $out/clk -- python3 scripts/conv_bench.py \
  1024,256,256,64,64 1024,128,128,128,128 2048,32,32,512,512 1024,64,64,256,256 512,256,256,64,64,0,1 2048,32,32,512,512,0,1 > $out/clk.log 2>&1
python3 scripts/clock_from_pmc.py $out/clk > gpurun_out/${tag}_clock_long.txt
python3 scripts/pmc_agg.py "$out/clk/**/*counter_collection.csv" >> gpurun_out/${tag}_clock_long.txt
grep -v amdgpu.ids $out/clk.log >> gpurun_out/${tag}_clock_long.txt
rm -rf $out
