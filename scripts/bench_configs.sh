#!/bin/bash
# The configs of BASELINE.json beside the headline one (C3 is `python bench.py`): for each, the bench line (with the CPU-oracle baseline of
# ITS model on the box's host cores and the roofline object) and the per-kernel table of its plan.
# usage (GPU box, repo root): bash scripts/bench_configs.sh <tag>     -> gpurun_out/<tag>_other_configs.txt, <tag>_<config>_kernel_table.txt
tag=${1:-cfg}
out=gpurun_out/${tag}_other_configs.txt
: > $out
run() {
  name=$1; shift
  echo "## $name: bench.py $*" >> $out
  python bench.py --steps 10 --warmup 3 --kernel-table --table-rows 30 "$@" > gpurun_out/${tag}_${name}.json 2> gpurun_out/${tag}_${name}_kernel_table.txt || { echo "FAILED" >> $out; return; }
  python - gpurun_out/${tag}_${name}.json >> $out <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
r, c = d.get("roofline", {}), d.get("cpu_baseline")
print(f"{d['value']} {d['unit']}  {d['ms_per_step']} ms/step  {d['dtype']}  | dominant {r.get('kernel')} frac {r.get('frac')} "
      f"(share {r.get('share_of_plan_time')}) | cpu {c['value'] if c else None} img/s on {c['cores'] if c else '-'} cores ({c['sample'].split(',')[-3].strip() if c else ''})")
PY
}
run C4_R2AttU_Net --model R2AttU_Net --batch 16 --dtype bf16
run C2_ResNetUnet_fp32 --model ResNetUnet --batch 32 --dtype fp32
run C2_ResNetUnet_bf16 --model ResNetUnet --batch 32 --dtype bf16 --no-cpu-baseline
run C5_vgg16_bn --model vgg16_bn --batch 16 --size 512 --dtype fp16
run C5_AttentionUNet512 --model AttentionUNet --batch 16 --size 512 --dtype fp16
run C1_resnet18 --model resnet18 --batch 8 --dtype fp32
run C3_fp16 --model AttentionUNet --batch 32 --dtype fp16 --no-cpu-baseline
cat $out
