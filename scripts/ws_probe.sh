#!/bin/bash
# Per-layer timing of the conv launches at HEAD (-> r04h_conv_layers.txt) and the timing-only builds of the weight-stationary
# kernel (what bounds it: shifted-column fragment reads, the DMA stream, the stores).
R=$PWD
python scripts/conv_layers.py > gpurun_out/r04h_conv_layers.txt 2>&1
for so in ws_noshift ws_nodma ws_nostore; do
  MI355_LIB=$R/ab/$so.so python scripts/conv_layers.py 2>/dev/null | grep conv3x3_ws > gpurun_out/r04h_ws_$so.txt
done
grep conv3x3_ws gpurun_out/r04h_conv_layers.txt | awk '{print $1, $2, $(NF-1)}' > /tmp/base.txt
for so in ws_noshift ws_nodma ws_nostore; do awk '{print $(NF-1)}' gpurun_out/r04h_ws_$so.txt > /tmp/$so.txt; done
echo "idx kernel base noshift nodma nostore"
paste -d' ' /tmp/base.txt /tmp/ws_noshift.txt /tmp/ws_nodma.txt /tmp/ws_nostore.txt
