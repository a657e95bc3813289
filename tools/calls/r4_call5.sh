# round 4, call 5: the whole GPU suite with the new tests (reference stage fixtures through the kernels, full-width float triangulation at
# three widths, world-size-2 dry run, tightened tolerances), the prefill GEMM tuning table, the XCD-hierarchical barrier probe, the lines
mkdir -p gpurun_out
timeout 2400 python -m pytest tests -m gpu -q -x --durations=15 > gpurun_out/r04e_pytest_gpu.txt 2>&1; tail -40 gpurun_out/r04e_pytest_gpu.txt
hipcc --offload-arch=gfx950 -O3 tools/probe/xcd_barrier_probe.hip -o /tmp/xb && timeout 300 /tmp/xb > gpurun_out/r04e_xcd_barrier_probe.txt 2>&1; cat gpurun_out/r04e_xcd_barrier_probe.txt
timeout 900 python tools/tune_prefill.py gpurun_out/prefill_gemms_gfx950.csv > gpurun_out/r04e_tune_prefill.txt 2>&1; tail -40 gpurun_out/r04e_tune_prefill.txt; ls -la gpurun_out/prefill_gemms_gfx950.csv; head -12 gpurun_out/prefill_gemms_gfx950.csv
bash tools/sweep.sh > gpurun_out/r04e_sweep.txt 2>&1 <<'S'
llava_auto||
llava_rb4||--wide-row-blocks 4
llava_rb8||--wide-row-blocks 8
qwen_auto||--model qwen7b
qwenfp8_auto||--model qwen7b-fp8
llava13b_auto||--model llava13b
llava_T1_auto||--temperature 1.0
S
cat gpurun_out/r04e_sweep.txt
