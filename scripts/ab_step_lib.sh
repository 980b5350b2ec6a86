for r in 1 2 3; do
  echo "prev $(MI355_LIB=ab/prev.so python bench.py --no-cpu-baseline | grep -o '"ms_per_step": [0-9.]*')"
  echo "new  $(python bench.py --no-cpu-baseline | grep -o '"ms_per_step": [0-9.]*')"
done
