# A/B of qknorm_rope_rows_kernel's head slots per wave (AHA_ROPE_CHUNK=8|10) inside the cfg 3 prefill (rocprofv3 averages)
cd ${GRAFT_REPO_ROOT:-/root/repo}
for v in 10 8 10 8; do echo "== chunk $v"; rm -rf gpurun_out/prof_rope; AHA_ROPE_CHUNK=$v bash scripts/prof_kernels.sh rope bench.py --steps 2 --warmup 1 --no-cpu-baseline 2>&1 | grep -E "qknorm_rope|vit_rope_pack" | cut -c1-130; done
