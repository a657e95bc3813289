# round 6, call 7: the whole GPU suite at the head, the driver's bench command, rocprofv3 summaries (one cohort-8 lane alone; the timed
# configuration), FETCH_SIZE / SQ counter passes of one cohort-8 lane (own --pmc runs, kernel-trace only)
mkdir -p gpurun_out
timeout 3000 python -m pytest tests/ -x -q -m gpu 2>&1 | tail -8 > gpurun_out/r06_gpu_suite.txt; cat gpurun_out/r06_gpu_suite.txt
( time python bench.py > gpurun_out/r06_bench_default.json 2> gpurun_out/r06_bench_default.err ) 2> gpurun_out/r06_bench_default.time; tail -3 gpurun_out/r06_bench_default.time; cut -c1-600 gpurun_out/r06_bench_default.json
bash tools/profile_bench.sh r06_1lane_cohort8 --lanes 1 --cohort 8; python tools/stats_summary.py gpurun_out/kernel_stats_r06_1lane_cohort8.csv 14
bash tools/profile_bench.sh r06_4lanes_cohort8; python tools/stats_summary.py gpurun_out/kernel_stats_r06_4lanes_cohort8.csv 14
bash tools/pmc_traffic.sh r06_fetch_size --lanes 1 --cohort 8 --no-vision-in-loop > gpurun_out/r06_pmc_fetch.log 2>&1; tail -25 gpurun_out/r06_pmc_fetch.log
C="SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE"
bash tools/pmc_counters.sh r06_sq_cohort8 "$C" --lanes 1 --cohort 8 --no-vision-in-loop > /dev/null 2>&1; ls -la gpurun_out/pmc_r06_sq_cohort8.json
