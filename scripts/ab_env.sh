#!/bin/bash
# Same-box A/B of one environment switch: scripts/ab_env.sh VAR "bench.py args" [rounds]
# Alternates VAR=0 / VAR=1 runs of bench.py (no CPU leg) and prints ms/step of each.
var=$1; args=$2; rounds=${3:-3}
for r in $(seq $rounds); do
  for v in 0 1; do
    ms=$(env $var=$v python bench.py $args --no-cpu-baseline | grep -o '"ms_per_step": [0-9.]*')
    echo "$var=$v $ms"
  done
done
