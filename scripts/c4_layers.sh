#!/bin/bash
# C4 (R2AttU_Net 256^2 bs 16): per-layer conv table aggregated by (kernel, shape), and the step under variant-threshold knobs
python scripts/conv_layers.py 16 r2attunet 256 2>/dev/null > gpurun_out/r04h_c4_conv_layers.txt
awk 'NR>1 {k=$2" "$3" "$4" "$5" "$6" "$7" "$8" "$9; n[k]++; ms[k]+=$(NF-1); tf[k]+=$NF} END {for (k in n) printf "%-90s n=%3d total %.3f ms avg %.4f ms %7.1f TFLOP/s\n", k, n[k], ms[k], ms[k]/n[k], tf[k]/n[k]}' gpurun_out/r04h_c4_conv_layers.txt | sort -t= -k2 -n -r > gpurun_out/r04h_c4_conv_groups.txt
cat gpurun_out/r04h_c4_conv_groups.txt
BENCH_ARGS="--model R2AttU_Net --batch 16" bash scripts/env_sweep.sh r04h_c4_knobs "" "MI355_WS128_TILE_MULT=3" "MI355_WS128_TILE_MULT=2" "MI355_PP128_FILL=50"
