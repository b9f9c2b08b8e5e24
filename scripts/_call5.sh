for abl in 0 4 5 0 4 5; do echo "== ABL=$abl"; AHA_GEMM_ABL=$abl timeout 120 python scripts/bench_gemm_data.py 2>&1 | grep -E "zeros|weights"; done
