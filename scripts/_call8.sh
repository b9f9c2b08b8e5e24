timeout 300 bash scripts/prof_kernels.sh cfg3 bench.py --steps 8 --warmup 2 --no-cpu-baseline 2>&1 | head -40
