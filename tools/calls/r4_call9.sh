# round 4, call 9: the fp8-activation (W8A8) path — MFMA operand layout probe, unit / bit-identity / loop tests, the bench line
mkdir -p gpurun_out
hipcc --offload-arch=gfx950 -O3 tools/probe/f8f6f4_probe.hip -o /tmp/f8 2>/dev/null && /tmp/f8 | tee gpurun_out/r04_f8f6f4_probe.txt
timeout 1500 python -m pytest tests/test_fp8a8_gpu.py -q -x > gpurun_out/r04h_pytest_fp8a8.txt 2>&1; tail -25 gpurun_out/r04h_pytest_fp8a8.txt | cut -c1-220
bash tools/sweep.sh > gpurun_out/r04h_sweep.txt 2>&1 <<'S'
qwenfp8||--model qwen7b-fp8
qwenfp8a8||--model qwen7b-fp8a8
S
cat gpurun_out/r04h_sweep.txt; tail -3 gpurun_out/sw_qwenfp8a8.err
