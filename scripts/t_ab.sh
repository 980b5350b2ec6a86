for round in 1 2 3; do for so in ab/cur.so ab/adamw_nt.so; do
  MI355_LIB=$PWD/$so python bench.py --no-cpu-baseline --no-profile --steps 20 --warmup 5 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$so', d['ms_per_step'])"
done; done
