# round 4, call 21: the round's head — whole GPU suite, smoke, the driver's bench command, kernel stats of one cohort lane and of the default line
mkdir -p gpurun_out
timeout 2400 python -m pytest tests -m gpu -q > gpurun_out/r04t_pytest_gpu.txt 2>&1; tail -4 gpurun_out/r04t_pytest_gpu.txt | cut -c1-200
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r04t_smoke.txt 2>&1; tail -1 gpurun_out/r04t_smoke.txt | cut -c1-200
timeout 900 python bench.py > gpurun_out/r04t_bench.json 2> gpurun_out/r04t_bench.err; cut -c1-400 gpurun_out/r04t_bench.json
bash tools/profile_bench.sh r04t_1lane --lanes 1 --cohort 4 --wide-row-blocks 84
python tools/stats_summary.py gpurun_out/kernel_stats_r04t_1lane.csv 8
