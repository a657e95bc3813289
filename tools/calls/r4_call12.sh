# round 4, call 12: (a) K / grid sweep of the wide kernel in the three formats (why fp8 gains nothing), (b) the whole GPU suite + smoke + the
# driver's bench command at the head that holds the W8A8 instantiations (every GEMM kernel's template / signature changed)
mkdir -p gpurun_out
timeout 600 python tools/fp8_k_sweep.py > gpurun_out/r04_fp8_k_sweep.txt 2>&1; tail -30 gpurun_out/r04_fp8_k_sweep.txt | cut -c1-200
timeout 2400 python -m pytest tests -m gpu -q > gpurun_out/r04k_pytest_gpu.txt 2>&1; tail -5 gpurun_out/r04k_pytest_gpu.txt | cut -c1-200
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r04k_smoke.txt 2>&1; tail -2 gpurun_out/r04k_smoke.txt | cut -c1-200
timeout 900 python bench.py > gpurun_out/r04k_bench.json 2> gpurun_out/r04k_bench.err; cut -c1-600 gpurun_out/r04k_bench.json
