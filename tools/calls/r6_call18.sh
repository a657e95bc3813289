timeout 900 python -m pytest tests/test_c8_gpu.py -x -q -m gpu -k "sampling_seeds" --tb=short 2>&1 | tail -15
