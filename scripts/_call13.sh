for div in 4 20 13 40 7; do echo "== PAGES_PER_BLOCK=$div"; AHA_ATTN_PAGES_PER_BLOCK=$div LENS=40960,41100,20000,131072 timeout 300 python scripts/bench_attn_decode.py 2>&1 | grep "L="; done
