#!/bin/bash
# The configs of BASELINE.json beside the headline one (C3 is `python bench.py`): one JSON line each into gpurun_out/configs_<tag>.txt
# usage (GPU box, repo root): bash scripts/bench_configs.sh <tag>
tag=${1:-cfg}
out=gpurun_out/configs_$tag.txt
: > $out
run() { echo "## $*" >> $out; python bench.py --no-cpu-baseline --no-profile --steps 10 --warmup 3 "$@" 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(d['value'], d['unit'], d['ms_per_step'], 'ms/step', d['dtype'])" >> $out; }
run --model R2AttU_Net --batch 16 --dtype bf16
run --model ResNetUnet --batch 32 --dtype fp32
run --model ResNetUnet --batch 32 --dtype bf16
run --model vgg16_bn --batch 16 --size 512 --dtype fp16
run --model resnet18 --batch 8 --dtype fp32
run --model AttentionUNet --batch 32 --dtype fp16
run --model AttentionUNet --batch 16 --size 512 --dtype fp16
cat $out
