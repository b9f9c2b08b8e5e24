# A/B of qknorm_rope_rows_kernel's head slots per wave (AHA_ROPE_CHUNK) and V-blocks-first grid order (AHA_ROPE_VFIRST) inside the cfg 3 prefill
cd ${GRAFT_REPO_ROOT:-/root/repo}
for v in "10 0" "5 0" "8 0" "10 1" "5 1" "10 0"; do set -- $v; echo "== chunk $1 vfirst $2"; rm -rf gpurun_out/prof_rope; AHA_ROPE_CHUNK=$1 AHA_ROPE_VFIRST=$2 bash scripts/prof_kernels.sh rope bench.py --steps 2 --warmup 1 --no-cpu-baseline 2>&1 | grep -E "qknorm_rope|vit_rope_pack|attn_prefill_kernel<128" | cut -c1-130; done
