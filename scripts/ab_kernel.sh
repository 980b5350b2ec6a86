# ab_kernel.sh "<kernel tag regex>" ab/A.so ab/B.so ...: per-layer rows of conv_layers.py for each library build (MI355_LIB)
pat="$1"; shift
for so in "$@"; do
  MI355_LIB=$PWD/$so python scripts/conv_layers.py 2>/dev/null | grep -E "$pat" > gpurun_out/ab_$(basename $so .so).txt
done
