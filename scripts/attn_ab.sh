# A/B of the prefill attention's score-chain variants (AHA_ATTN_SMX) and the static wave priority (AHA_ATTN_TUNE): the stand-alone
# text kernel (scripts/bench_attn.py) and the cfg 3 prefill's kernels under rocprofv3 (ViT 96/80 and text 128/128 instantiations)
cd ${GRAFT_REPO_ROOT:-/root/repo}
for v in "0 0" "1 0" "2 0" "1 1" "0 1" "0 0" "1 0"; do set -- $v; echo "== smx $1 tune $2"; AHA_ATTN_SMX=$1 AHA_ATTN_TUNE=$2 timeout 300 python scripts/bench_attn.py 2>&1 | grep ms/launch; AHA_ATTN_SMX=$1 AHA_ATTN_TUNE=$2 timeout 300 python scripts/bench_attn.py 1542 2>&1 | grep ms/launch; done
for v in "0 0" "1 0" "2 0" "1 1" "0 0" "1 0"; do set -- $v; echo "== rocprof bench smx $1 tune $2"; rm -rf gpurun_out/prof_ab; AHA_ATTN_SMX=$1 AHA_ATTN_TUNE=$2 bash scripts/prof_kernels.sh ab bench.py --steps 8 --warmup 2 2>&1 | grep -E "attn_prefill|gemm256q_kernel<0" | cut -c1-130; done
