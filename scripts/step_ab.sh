for round in 1 2 3; do for so in "$@"; do
  v=$(MI355_LIB=$PWD/$so python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-profile 2>/dev/null | python -c "import json,sys; print(json.loads(sys.stdin.read())['ms_per_step'])")
  echo "$so $v"
done; done
