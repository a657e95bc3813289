# round 6, call 2: HBM activity (driver counters) of the timed configuration next to a calibration stream; per-workgroup clocks incl. the
# draft's slab GEMMs / reduces / batched small kernels; the c8 per-element bars with the fp32-chain floor; the new bench line (short)
mkdir -p gpurun_out
timeout 900 python tools/mem_activity.py gpurun_out/r06_mem_activity_4lanes_cohort8.json --steps 3 --warmup 1 --no-cpu-baseline --no-ar > gpurun_out/r06_mem_activity.log 2>&1; tail -60 gpurun_out/r06_mem_activity.log
export VISPEC_LIB_VARIANT=wgclk
timeout 900 python tools/wg_clock.py --lanes 4 --cohort 8 --no-vision --max-new-tokens 256 --cap 100000000 gpurun_out/r06_wgclock_4lanes_cohort8_all.json > gpurun_out/r06_wgclock_4lanes_all.log 2>&1; tail -3 gpurun_out/r06_wgclock_4lanes_all.log
timeout 900 python tools/wg_clock.py --lanes 1 --cohort 8 --no-vision --max-new-tokens 256 gpurun_out/r06_wgclock_1lane_cohort8_all.json > gpurun_out/r06_wgclock_1lane_all.log 2>&1; tail -3 gpurun_out/r06_wgclock_1lane_all.log
unset VISPEC_LIB_VARIANT
timeout 1500 python -m pytest tests/test_c8_gpu.py -q -m gpu -k "fp64_product" --tb=line 2>&1 | tail -15 > gpurun_out/r06_c8_tests.txt; cat gpurun_out/r06_c8_tests.txt
timeout 900 python bench.py --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/r06_bench_short.json 2> gpurun_out/r06_bench_short.err; tail -c 3000 gpurun_out/r06_bench_short.json; tail -5 gpurun_out/r06_bench_short.err
