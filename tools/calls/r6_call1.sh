# round 6, call 1: per-workgroup clocks of the timed configuration (4 lanes x cohort 8) and of one lane alone (diagnostic build
# libvispec_hip_wgclk.so, csrc/wgclock.h); the tightened c8 GEMM bars and the new unstructured-weights tests
mkdir -p gpurun_out
export VISPEC_LIB_VARIANT=wgclk
timeout 900 python tools/wg_clock.py --lanes 4 --cohort 8 gpurun_out/r06_wgclock_4lanes_cohort8.json > gpurun_out/r06_wgclock_4lanes.log 2>&1; tail -5 gpurun_out/r06_wgclock_4lanes.log
timeout 900 python tools/wg_clock.py --lanes 1 --cohort 8 gpurun_out/r06_wgclock_1lane_cohort8.json > gpurun_out/r06_wgclock_1lane.log 2>&1; tail -5 gpurun_out/r06_wgclock_1lane.log
timeout 900 python tools/wg_clock.py --lanes 4 --cohort 8 --no-vision --max-new-tokens 400 gpurun_out/r06_wgclock_4lanes_cohort8_prefill_light.json > gpurun_out/r06_wgclock_4lanes_pl.log 2>&1; tail -5 gpurun_out/r06_wgclock_4lanes_pl.log
unset VISPEC_LIB_VARIANT
timeout 1500 python -m pytest tests/test_unstructured_gpu.py -x -q -m gpu -s 2>&1 | tail -40 > gpurun_out/r06_unstructured_tests.txt; cat gpurun_out/r06_unstructured_tests.txt
timeout 1500 python -m pytest tests/test_c8_gpu.py -q -m gpu -k "fp64_product" 2>&1 | tail -15 > gpurun_out/r06_c8_tests.txt; cat gpurun_out/r06_c8_tests.txt
