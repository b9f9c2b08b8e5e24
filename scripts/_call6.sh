( timeout 300 python -m pytest tests/test_ops_gpu.py -k "gemm" -x -q ) 2>&1 | tail -2
timeout 300 python scripts/tune_gemm.py 2>&1 | grep "M="
