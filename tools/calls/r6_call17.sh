timeout 900 python -m pytest tests/test_cohort_gpu.py -x -q -m gpu -k "wide_trees or qwen_tiny" --tb=short 2>&1 | tail -15
