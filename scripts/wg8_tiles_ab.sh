#!/bin/bash
# eight-wave weight gradient: square tile blocks per XCD (HEAD) against row-major tiles (ab/rowtiles.so): parity subset, step, PMC fetch
R=$PWD
python -m pytest tests/test_gpu_conv.py tests/test_gpu_bench_scale.py -x -q -k "wgrad or weight_grad or gradient or scale" 2>&1 | tail -2 || exit 1
bash scripts/env_sweep.sh r04j_wg8_tiles "" "MI355_LIB=$R/ab/rowtiles.so"
export TMPDIR=/tmp
out=$R/gpurun_out/r04j_pmc
rm -rf $out; mkdir -p $out
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $out/fetch -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-profile > $out/fetch.log 2>&1
python3 scripts/pmc_agg.py "$out/fetch/**/*counter_collection.csv" > gpurun_out/r04j_pmc_fetch_size.txt
rm -rf $out
grep -i "wgrad" gpurun_out/r04j_pmc_fetch_size.txt
