#!/bin/bash
# SQ-side counters of the current build (run on the GPU box from the repo root): scripts/profile_sq.sh r01f
#   <tag>_pmc_sq_counters.txt : per-kernel sums of the SQ counters (quad-cycle units except MFMA_BUSY, which counts cycles:
#                               16 per v_mfma_f32_16x16x32_bf16)
#   <tag>_pmc_clock.txt       : effective clock per kernel = GRBM_GUI_ACTIVE / 8 / duration (MI355X_MICROARCH.md, DVFS)
set -e -o pipefail
tag=$1
out=$PWD/gpurun_out/$tag
mkdir -p $out
export TMPDIR=/tmp
B="python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-profile --serial-streams"
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_BUSY_CYCLES --output-format csv -d $out/pmc_sq -- $B > $out/pmc_sq.log 2>&1
python3 scripts/pmc_agg.py "$out/pmc_sq/**/*counter_collection.csv" > $out/${tag}_pmc_sq_counters.txt
echo "[profile] sq pass done"
rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE --output-format csv -d $out/pmc_clk -- $B > $out/pmc_clk.log 2>&1
python3 scripts/clock_from_pmc.py $out/pmc_clk > $out/${tag}_pmc_clock.txt
rm -rf $out/pmc_sq $out/pmc_clk
cat $out/${tag}_pmc_clock.txt
cut -c1-200 $out/${tag}_pmc_sq_counters.txt | head -8
