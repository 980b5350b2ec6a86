# ab_kernel.sh "<kernel tag regex>" ab/A.so ab/B.so ...: per-layer rows of conv_layers.py for each library build (MI355_LIB);
# extra environment for the runs through AB_ENV="K=V K=V"
pat="$1"; shift
for so in "$@"; do
  env $AB_ENV MI355_LIB=$PWD/$so python scripts/conv_layers.py 2>/dev/null | grep -E "$pat" > gpurun_out/ab_$(basename $so .so).txt
done
