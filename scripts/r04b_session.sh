set -o pipefail
bash scripts/bench_configs.sh r04b > gpurun_out/r04b_configs_stdout.txt 2>&1
python tests/diag/diag_first_launch.py vgg16_bn 16 512 fp16 > gpurun_out/r04b_first_launch.txt 2>&1
python tests/diag/diag_first_launch.py attentionunet 32 256 bf16 >> gpurun_out/r04b_first_launch.txt 2>&1
R2_HELD=256 R2_DECAY_AT=32 R2_LR2=1e-4 python tests/diag/diag_r2_bf16_spread.py 32 40 48 56 64 > gpurun_out/r04b_r2_conditioned.txt 2>&1
cat gpurun_out/r04b_other_configs.txt; cat gpurun_out/r04b_first_launch.txt; cat gpurun_out/r04b_r2_conditioned.txt
