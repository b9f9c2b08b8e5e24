mkdir -p gpurun_out
( timeout 600 python -m pytest tests/test_ops_gpu.py -k "gemm or attn" -x -q ) > gpurun_out/t2.log 2>&1; echo "rc=$?" >> gpurun_out/t2.log
tail -5 gpurun_out/t2.log
echo "== QUAD=0"; AHA_GEMM_QUAD=0 timeout 120 python scripts/bench_gemm.py 2>&1 | grep TFLOP
echo "== QUAD=1"; AHA_GEMM_QUAD=1 timeout 120 python scripts/bench_gemm.py 2>&1 | grep TFLOP
echo "== data QUAD=1"; timeout 120 python scripts/bench_gemm_data.py 2>&1 | grep TFLOP
echo "== attn"; timeout 200 python scripts/bench_attn.py 2048 8192 40980 2>&1 | tail -8
