for sk in 0 1 2; do for sh in 3 8; do echo "== SKEW=$sk SHIFT=$sh"; AHA_ATTN_SKEW=$sk AHA_ATTN_SKEW_SHIFT=$sh timeout 200 python scripts/bench_attn.py 8192 40980 2>&1 | grep attn_prefill; done; done
