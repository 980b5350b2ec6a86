#!/bin/bash
# The per-round profile set committed under profiles/ (run on the GPU box from the repo root):
#   scripts/profile_round.sh r01f
# 1. two PMC passes (FETCH_SIZE, WRITE_SIZE; --kernel-trace only, as the pool requires) -> <tag>_pmc_traffic.json
# 2. the default bench with its HIP-event kernel table (reads the traffic file written in 1)
# 3. rocprofv3 --kernel-trace --stats of the single-stream replay (what bench.py's launch timing measures) and of the
#    production schedule (weight-gradient kernels on a side stream)
set -e -o pipefail
tag=$1
out=$PWD/gpurun_out/$tag
mkdir -p $out
export TMPDIR=/tmp
B="python3 bench.py"
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $out/pmc_fetch -- $B --steps 2 --warmup 1 --no-cpu-baseline --no-profile > $out/pmc_fetch.log 2>&1
echo "[profile] fetch pass done"
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $out/pmc_write -- $B --steps 2 --warmup 1 --no-cpu-baseline --no-profile > $out/pmc_write.log 2>&1
echo "[profile] write pass done"
python3 scripts/pmc_traffic.py "$out/pmc_fetch/**/*counter_collection.csv" "$out/pmc_write/**/*counter_collection.csv" profiles/${tag}_pmc_traffic.json
python3 scripts/pmc_agg.py "$out/pmc_fetch/**/*counter_collection.csv" > $out/${tag}_pmc_fetch_size.txt
python3 scripts/pmc_agg.py "$out/pmc_write/**/*counter_collection.csv" > $out/${tag}_pmc_write_size.txt
cp profiles/${tag}_pmc_traffic.json $out/
$B --steps 20 --warmup 5 --kernel-table > $out/${tag}_bench_n1_bf16.json 2> $out/${tag}_bench_n1_bf16_kernel_table.txt
echo "[profile] bench done: $(cut -c1-120 $out/${tag}_bench_n1_bf16.json)"
rocprofv3 --kernel-trace --stats --output-format csv -d $out/serial -- $B --steps 10 --warmup 3 --no-cpu-baseline --serial-streams > $out/${tag}_bench_n1_bf16_serial_profiled.json 2> $out/serial.log
cp $(find $out/serial -name "*kernel_stats.csv" | head -1) $out/${tag}_bench_n1_bf16_serial_kernel_stats.csv
echo "[profile] serial rocprof done"
rocprofv3 --kernel-trace --stats --output-format csv -d $out/prod -- $B --steps 10 --warmup 3 --no-cpu-baseline > $out/${tag}_bench_n1_bf16_profiled.json 2> $out/prod.log
cp $(find $out/prod -name "*kernel_stats.csv" | head -1) $out/${tag}_bench_n1_bf16_kernel_stats.csv
rm -rf $out/pmc_fetch $out/pmc_write $out/serial $out/prod
ls -la $out
