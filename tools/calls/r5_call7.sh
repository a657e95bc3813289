mkdir -p gpurun_out
timeout 3000 python -m pytest tests/test_full_size_gpu.py -q -m gpu -s -k "full_width" 2>&1 | grep -v "^$" | grep "full-width\|passed\|failed\|FAILED\|Error" > gpurun_out/r05_fullwidth_tests.txt; cat gpurun_out/r05_fullwidth_tests.txt
