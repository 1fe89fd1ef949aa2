timeout 900 python -m pytest tests/test_gpu_ops.py -x -q -m gpu -k noise 2>&1 | grep -B3 -A10 "Error" | cut -c1-300 | head -40
