out=gpurun_out/r04d_ws_order.txt; : > $out
L="32,256,256,64,64 32,128,128,64,128 32,256,256,64,128 32,256,256,128,64 32,128,128,128,128 32,128,128,128,256"
for rep in 1 2; do
  for lib in ab/ws_roworder.so ""; do
    echo "== ${lib:-default (column order)}" >> $out
    MI355_LIB=${lib:+$PWD/$lib} python scripts/conv_bench.py $L 2>/dev/null | awk '{print $1, $2, $4, $7, "ms", $(NF-5), "TFLOP/s"}' >> $out
  done
done
cat $out
scripts/env_sweep.sh r04d_ws_order_step "" "MI355_LIB=$PWD/ab/ws_roworder.so"
