( timeout 600 python -m pytest tests/test_ops_gpu.py tests/test_vl_gpu.py -k "gemm or attn or vl" -x -q ) 2>&1 | tail -2
timeout 300 bash scripts/prof_kernels.sh cfg3 bench.py --steps 8 --warmup 2 --no-cpu-baseline 2>&1 | head -24
grep -o '"prefill_ms": [0-9.]*' gpurun_out/prof_cfg3/run.log
timeout 200 python scripts/bench_attn.py 2048 8192 40980 2>&1 | grep attn_prefill
