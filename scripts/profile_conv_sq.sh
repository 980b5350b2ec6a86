#!/bin/bash
# SQ counters of single conv launches (scripts/conv_bench.py shapes) for one library build: scripts/profile_conv_sq.sh TAG LIB.so shape...
# writes gpurun_out/TAG_sq.txt (per-kernel sums; SQ_* are summed over the shader engines' samples, MFMA_BUSY counts pipe cycles)
set -e -o pipefail
tag=$1; so=$2; shift 2
out=$PWD/gpurun_out/prof_$tag
mkdir -p $out
export TMPDIR=/tmp
export MI355_LIB=$so
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_BUSY_CYCLES --output-format csv -d $out/sq -- python3 scripts/conv_bench.py "$@" > $out/sq.log 2>&1
python3 scripts/pmc_agg.py "$out/sq/**/*counter_collection.csv" > gpurun_out/${tag}_sq.txt
rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE --output-format csv -d $out/clk -- python3 scripts/conv_bench.py "$@" > $out/clk.log 2>&1
python3 scripts/clock_from_pmc.py $out/clk > gpurun_out/${tag}_clock.txt || true
rm -rf $out
