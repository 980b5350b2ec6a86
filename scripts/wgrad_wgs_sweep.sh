for r in 1 2; do for w in 256 320 384 512; do echo "WGS=$w $(MI355_WGRAD_WGS=$w python bench.py --no-cpu-baseline | grep -o '"ms_per_step": [0-9.]*')"; done; done
