( AHA_GEMM_STAGE=1 timeout 300 python -m pytest tests/test_ops_gpu.py -k "gemm_plain or every_tile" -x -q ) 2>&1 | tail -3
echo "== STAGE=0"; AHA_GEMM_ONLY=qkv,o,big timeout 120 python scripts/bench_gemm.py 2>&1 | grep TFLOP
echo "== STAGE=1"; AHA_GEMM_STAGE=1 AHA_GEMM_ONLY=qkv,o,big timeout 120 python scripts/bench_gemm.py 2>&1 | grep TFLOP
echo "== data STAGE=0"; timeout 120 python scripts/bench_gemm_data.py 2>&1 | grep TFLOP
echo "== data STAGE=1"; AHA_GEMM_STAGE=1 timeout 120 python scripts/bench_gemm_data.py 2>&1 | grep TFLOP
