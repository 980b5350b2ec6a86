#!/bin/bash
# Same-box A/B of library builds over the whole step: scripts/ab_step_lib.sh ab/A.so ab/B.so ... [rounds=3]
# ("default" = the in-tree library); prints ms/step of bench.py --no-cpu-baseline, alternating the builds
rounds=3
libs=()
for a in "$@"; do if [[ "$a" =~ ^[0-9]+$ ]]; then rounds=$a; else libs+=("$a"); fi; done
for r in $(seq $rounds); do
  for so in default "${libs[@]}"; do
    if [ "$so" = default ]; then ms=$(python bench.py --no-cpu-baseline | grep -o '"ms_per_step": [0-9.]*'); else ms=$(MI355_LIB=$so python bench.py --no-cpu-baseline | grep -o '"ms_per_step": [0-9.]*'); fi
    echo "$so $ms"
  done
done
