( timeout 900 python -m pytest tests/test_ops_gpu.py -k "four_wave" -x -q ) 2>&1 | tail -4
