python -m pytest tests/test_gpu_c5.py -m gpu -q -x 2>&1 | tail -15
