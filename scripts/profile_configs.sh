#!/bin/bash
# rocprofv3 --kernel-trace --stats of the single-stream replay of the other configs' plans at HEAD (what bench.py's HIP-event kernel
# tables measure): scripts/profile_configs.sh <tag>  ->  gpurun_out/<tag>_<config>_serial_kernel_stats.csv
tag=$1
export TMPDIR=/tmp
run() {
  name=$1; shift
  out=$PWD/gpurun_out/${tag}_prof_$name
  rm -rf $out; mkdir -p $out
  rocprofv3 --kernel-trace --stats --output-format csv -d $out -- python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-profile --serial-streams "$@" > $out/bench.json 2> $out/log.txt || { echo "$name FAILED"; tail -3 $out/log.txt; return; }
  cp $(find $out -name "*kernel_stats.csv" | head -1) gpurun_out/${tag}_${name}_serial_kernel_stats.csv
  cut -c1-160 $out/bench.json
  head -6 gpurun_out/${tag}_${name}_serial_kernel_stats.csv | cut -c1-150
  rm -rf $out
}
run C4_R2AttU_Net --model R2AttU_Net --batch 16 --dtype bf16
run C2_ResNetUnet_bf16 --model ResNetUnet --batch 32 --dtype bf16
run C5_vgg16_bn --model vgg16_bn --batch 16 --size 512 --dtype fp16
run C5_AttentionUNet512 --model AttentionUNet --batch 16 --size 512 --dtype fp16
