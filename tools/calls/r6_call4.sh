# round 6, call 4: CU-mask layouts (which one pins a lane to its XCDs?) with the XCC ids of the workgroups; trees of 33..64 nodes in cohorts;
# world-8 dry run; a full-size checkpoint through from_pretrained; regression of the cohort / loop suites after the two-tile views
mkdir -p gpurun_out
VISPEC_LIB_VARIANT=wgclk timeout 900 python tools/cu_mask_probe.py 4 > gpurun_out/r06_cu_mask_probe_wgclk.txt 2>&1; tail -5 gpurun_out/r06_cu_mask_probe_wgclk.txt | cut -c1-1500
timeout 900 python tools/cu_mask_probe.py 4 > gpurun_out/r06_cu_mask_probe.txt 2>&1; tail -5 gpurun_out/r06_cu_mask_probe.txt
timeout 1500 python -m pytest tests/test_c8_gpu.py -x -q -m gpu -k "more_than_one_tile or at_most_four or whole_loops or eight_slots or tree_shapes" --tb=short 2>&1 | tail -25 > gpurun_out/r06_wide_tree_tests.txt; cat gpurun_out/r06_wide_tree_tests.txt
timeout 1700 python -m pytest tests/test_world8_gpu.py -x -q -m gpu -s 2>&1 | tail -8 > gpurun_out/r06_world8_test.txt; cat gpurun_out/r06_world8_test.txt
timeout 2400 python -m pytest tests/test_checkpoint_full_size_gpu.py -x -q -m gpu -s --tb=short 2>&1 | tail -25 > gpurun_out/r06_checkpoint_full_size.txt; cat gpurun_out/r06_checkpoint_full_size.txt
timeout 2400 python -m pytest tests/test_cohort_gpu.py tests/test_wide_gpu.py tests/test_loop_gpu.py tests/test_fp8a8_gpu.py -x -q -m gpu 2>&1 | tail -5
