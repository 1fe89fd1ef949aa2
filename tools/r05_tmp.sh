timeout 900 python -m pytest tests/test_gpu_ops.py -x -q -m gpu 2>&1 | tail -12 | cut -c1-250
