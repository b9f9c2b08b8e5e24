( timeout 600 python -m pytest tests/test_ops_gpu.py tests/test_vl_gpu.py -k "gemm or vl" -x -q ) 2>&1 | tail -2
timeout 300 bash scripts/prof_kernels.sh cfg3 bench.py --steps 8 --warmup 2 --no-cpu-baseline 2>&1 | grep -E "gemm|attn_prefill"
grep -o '"prefill_ms": [0-9.]*' gpurun_out/prof_cfg3/run.log
