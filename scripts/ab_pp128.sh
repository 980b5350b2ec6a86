for so in "$@"; do
  MI355_LIB=$PWD/$so python scripts/conv_layers.py 2>/dev/null | grep pp128 > gpurun_out/ab_$(basename $so .so).txt
done
