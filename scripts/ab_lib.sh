#!/bin/bash
# A/B builds of libmi355conv.so on the same box: ab_lib.sh "<kernel tag regex>" ab/A.so ab/B.so ...  (two interleaved rounds;
# MI355_LIB selects the library, see mi355/lib.py)
pat="$1"; shift
for round in 1 2; do
  for so in "$@"; do
    MI355_LIB=$PWD/$so python scripts/conv_layers.py 2>/dev/null | awk -v t=$so -v pat="$pat" '$0 ~ pat {ms+=$(NF-1)} END {printf "%-28s total ms %.3f\n", t, ms}'
  done
done
