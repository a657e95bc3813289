# round 6, call 16: W8A8 cohorts with wide trees
timeout 900 python -m pytest tests/test_fp8a8_gpu.py -x -q -m gpu -k "wide_trees or cohort_of_four" --tb=short 2>&1 | tail -15
