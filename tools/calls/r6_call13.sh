# round 6, call 13: final head — the whole GPU suite, smoke, the no-flags bench line
mkdir -p gpurun_out
timeout 3000 python -m pytest tests/ -x -q -m gpu 2>&1 | tail -4 > gpurun_out/r06_gpu_suite_head.txt; cat gpurun_out/r06_gpu_suite_head.txt
python -c "import __graft_entry__ as g; g.smoke()"
( time python bench.py > gpurun_out/r06_bench_default.json 2> gpurun_out/r06_bench_default.err ) 2> gpurun_out/r06_bench_default.time; tail -3 gpurun_out/r06_bench_default.time; cut -c1-200 gpurun_out/r06_bench_default.json
