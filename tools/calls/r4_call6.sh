# round 4, call 6: whole GPU suite (all failures), barrier probe (every wave drains), recorded prefill GEMMs A/B, fp8 prefill epilogue, lines
mkdir -p gpurun_out
timeout 2700 python -m pytest tests -m gpu -q --durations=12 > gpurun_out/r04f_pytest_gpu.txt 2>&1; tail -45 gpurun_out/r04f_pytest_gpu.txt | cut -c1-220
hipcc --offload-arch=gfx950 -O3 tools/probe/xcd_barrier_probe.hip -o /tmp/xb && timeout 300 /tmp/xb > gpurun_out/r04f_xcd_barrier_probe.txt 2>&1; cat gpurun_out/r04f_xcd_barrier_probe.txt
timeout 600 python tools/prefill_breakdown.py > gpurun_out/r04f_prefill_llava_recorded.txt 2>&1; grep "^iter" gpurun_out/r04f_prefill_llava_recorded.txt
VISPEC_PREFILL_GEMMS=default timeout 600 python tools/prefill_breakdown.py > gpurun_out/r04f_prefill_llava_default.txt 2>&1; grep "^iter" gpurun_out/r04f_prefill_llava_default.txt
MODEL=qwen7b-fp8 timeout 600 python tools/prefill_breakdown.py > gpurun_out/r04f_prefill_fp8.txt 2>&1; grep "^iter" gpurun_out/r04f_prefill_fp8.txt
MODEL=qwen7b timeout 600 python tools/prefill_breakdown.py > gpurun_out/r04f_prefill_qwen.txt 2>&1; grep "^iter" gpurun_out/r04f_prefill_qwen.txt
bash tools/sweep.sh > gpurun_out/r04f_sweep.txt 2>&1 <<'S'
llava_auto||
llava_auto_defaultgemms|VISPEC_PREFILL_GEMMS=default|
llava_rb4||--wide-row-blocks 4
qwen_auto||--model qwen7b
qwenfp8_auto||--model qwen7b-fp8
llava13b_auto||--model llava13b
S
cat gpurun_out/r04f_sweep.txt
