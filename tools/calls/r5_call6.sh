# round 5, call 6: full-size cohort-8 tests (every model), full-width float parity of the cohort-8 verify forward (+ the new 13B width)
mkdir -p gpurun_out
timeout 3000 python -m pytest tests/test_full_size_gpu.py -x -q -m gpu -s -k "cohort_of_eight or default_bench or full_width" 2>&1 | grep -v "^$" | tail -40 > gpurun_out/r05_fullsize_c8_tests.txt; cat gpurun_out/r05_fullsize_c8_tests.txt
