# round 6, call 8: the whole GPU suite at the head (lstk_row<20> removed, test fixes), the rocprofv3 summary of the timed configuration,
# the driver's command with more steps
mkdir -p gpurun_out
timeout 3000 python -m pytest tests/ -x -q -m gpu 2>&1 | tail -6 > gpurun_out/r06_gpu_suite_head.txt; cat gpurun_out/r06_gpu_suite_head.txt
bash tools/profile_bench.sh r06_4lanes_cohort8 --lanes 4 --cohort 8; python tools/stats_summary.py gpurun_out/kernel_stats_r06_4lanes_cohort8.csv 16
( time python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r06_bench_steps20.json 2> gpurun_out/r06_bench_steps20.err ) 2> gpurun_out/r06_bench_steps20.time; tail -3 gpurun_out/r06_bench_steps20.time; cut -c1-300 gpurun_out/r06_bench_steps20.json
python -c "import __graft_entry__ as g; g.smoke()"
