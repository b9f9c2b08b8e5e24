# A/B of the prefill attention's row-order epilogue stores (AHA_ATTN_EPI_ROWS): stand-alone text kernel and the cfg 3 prefill's kernels
cd ${GRAFT_REPO_ROOT:-/root/repo}
timeout 600 python -m pytest tests/test_ops_gpu.py tests/test_vl_gpu.py tests/test_asr_gpu.py -q -x -k "attn or vl or asr" 2>&1 | grep -E "passed|failed"
for v in 1 0 1 0; do echo "== epi_rows $v"; AHA_ATTN_EPI_ROWS=$v timeout 300 python scripts/bench_attn.py 2>&1 | grep ms/launch; AHA_ATTN_EPI_ROWS=$v timeout 300 python scripts/bench_attn.py 1542 2>&1 | grep ms/launch; done
for v in 1 0 1 0; do echo "== rocprof bench epi_rows $v"; rm -rf gpurun_out/prof_ab; AHA_ATTN_EPI_ROWS=$v bash scripts/prof_kernels.sh ab bench.py --steps 4 --warmup 1 --no-cpu-baseline 2>&1 | grep -E "attn_prefill" | cut -c1-130; grep -o '"prefill_ms": [0-9.]*' gpurun_out/prof_ab/run.log; done
