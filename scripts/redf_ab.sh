#!/bin/bash
# rows in flight per thread in the BatchNorm backward apply pass (A/B builds), C3 and C4
R=$PWD
V=("" "MI355_LIB=$R/ab/applyf8.so" "MI355_LIB=$R/ab/applyf6.so" "MI355_LIB=$R/ab/applyf2.so")
bash scripts/env_sweep.sh r04i_applyf "${V[@]}"
BENCH_ARGS="--model R2AttU_Net --batch 16" bash scripts/env_sweep.sh r04i_applyf_c4 "${V[@]}"
