#!/bin/bash
# step time of the default bench under environment variants, alternating: env_sweep.sh TAG "A=1 B=2" "C=3" ...   ("" = defaults)
out=gpurun_out/$1.txt; shift
: > $out
step() { python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-profile ${BENCH_ARGS} 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'])"; }
for rep in 1 2 3; do
  for v in "$@"; do echo "== [${v:-defaults}]" >> $out; env $v bash -c "$(declare -f step); step" >> $out || exit 1; done
done
paste - - < $out
