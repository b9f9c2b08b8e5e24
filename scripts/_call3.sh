mkdir -p gpurun_out
( timeout 600 python -m pytest tests/test_ops_gpu.py -k "gemm" -x -q ) > gpurun_out/t3.log 2>&1; echo "rc=$?" >> gpurun_out/t3.log
tail -3 gpurun_out/t3.log
echo "== BAR2=1"; timeout 120 python scripts/bench_gemm.py 2>&1 | grep TFLOP
echo "== BAR2=0 (plain only)"; AHA_GEMM_BAR2=0 AHA_GEMM_ONLY=qkv,o,big timeout 120 python scripts/bench_gemm.py 2>&1 | grep TFLOP
echo "== BAR2=1 GROUP=4"; AHA_GEMM_GROUP=4 AHA_GEMM_ONLY=qkv,gateup,big timeout 120 python scripts/bench_gemm.py 2>&1 | grep TFLOP
echo "== data BAR2=1"; timeout 120 python scripts/bench_gemm_data.py 2>&1 | grep TFLOP
