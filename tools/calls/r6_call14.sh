# round 6, call 14: the wide-tree stream test
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_c8_gpu.py -x -q -m gpu -k "wide_trees or more_than_one_tile or at_most_four" --tb=short 2>&1 | tail -12
