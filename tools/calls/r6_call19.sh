# round 6, call 19: the whole GPU suite and smoke() at the final head
timeout 3000 python -m pytest tests/ -x -q -m gpu 2>&1 | tail -4 > gpurun_out/r06_gpu_suite_head.txt; cat gpurun_out/r06_gpu_suite_head.txt
python -c "import __graft_entry__ as g; g.smoke()"
