cd ${GRAFT_REPO_ROOT:-/root/repo}
timeout 900 python -m pytest tests/test_ops_gpu.py -q -x -k "gemm" 2>&1 | tail -3
timeout 900 python -m pytest tests/test_vl_gpu.py -q -x 2>&1 | tail -3
timeout 900 python -m pytest tests/test_baseline_fullsize_parity_gpu.py -q -x -k "vit or cfg3 or vision" 2>&1 | tail -3
python scripts/bench_gemm_fc2.py 2>&1 | tail -9
rm -rf gpurun_out/prof_fc2; bash scripts/prof_kernels.sh fc2 bench.py --steps 8 --warmup 2 30 2>&1 | cut -c1-130
grep -o '"prefill_ms": [0-9.]*\|"value": [0-9.]*' gpurun_out/prof_fc2/run.log | head
python bench.py --steps 20 --warmup 3 2>/dev/null | tail -1 | cut -c1-400
