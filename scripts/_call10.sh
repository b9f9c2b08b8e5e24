( timeout 900 python -m pytest tests/test_tp_gpu.py -x -q ) 2>&1 | tail -15
