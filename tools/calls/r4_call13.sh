# round 4, call 13: the eight-row-block form of the W8A8 cohort GEMM (gemm_w32_wide8_kernel<.., 2, ..>): parity, the fp8 bench line A/B on one box
# (W8A16 rb4 | W8A8 rb4 | W8A8 rb8 | W8A8 rb8 with three groups of lookahead), per-kernel times of the rb8 form on one lane
mkdir -p gpurun_out
hipcc --offload-arch=gfx950 -O3 tools/probe/f8f6f4_probe.hip -o /tmp/f8 2>/dev/null && /tmp/f8 | tee gpurun_out/r04_f8f6f4_probe.txt
timeout 1500 python -m pytest tests/test_fp8a8_gpu.py -q > gpurun_out/r04l_pytest_fp8a8.txt 2>&1; tail -6 gpurun_out/r04l_pytest_fp8a8.txt | cut -c1-250
timeout 400 python tools/fp8_k_sweep.py > gpurun_out/r04l_fp8_k_sweep_rb4.txt 2>&1; grep -A3 "n_req 4" gpurun_out/r04l_fp8_k_sweep_rb4.txt | cut -c1-200
RB=8 timeout 400 python tools/fp8_k_sweep.py > gpurun_out/r04l_fp8_k_sweep_rb8.txt 2>&1; grep -A3 "n_req 4" gpurun_out/r04l_fp8_k_sweep_rb8.txt | cut -c1-200
bash tools/sweep.sh > gpurun_out/r04l_sweep.txt 2>&1 <<'S'
l_fp8_a||--model qwen7b-fp8
l_a8_rb4_a||--model qwen7b-fp8a8 --wide-row-blocks 4
l_a8_rb8_a||--model qwen7b-fp8a8 --wide-row-blocks 8
l_fp8_b||--model qwen7b-fp8
l_a8_rb4_b||--model qwen7b-fp8a8 --wide-row-blocks 4
l_a8_rb8_b||--model qwen7b-fp8a8 --wide-row-blocks 8
S
cat gpurun_out/r04l_sweep.txt
bash tools/profile_bench.sh r04l_fp8a8_rb8_1lane --model qwen7b-fp8a8 --lanes 1 --cohort 4 --wide-row-blocks 8
python tools/stats_summary.py gpurun_out/kernel_stats_r04l_fp8a8_rb8_1lane.csv 12
# three groups of lookahead (12 KiB of W per wave in flight): rebuilt on the box, the default library restored afterwards
cp vispec_amd/libvispec_hip.so /tmp/lib_la2.so
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -Wno-unused-result -DVISPEC_WIDE8_A8_LA=3 vispec_amd/csrc/vispec_hip.hip -o vispec_amd/libvispec_hip.so 2> gpurun_out/r04l_build_la3.err
bash tools/sweep.sh > gpurun_out/r04l_sweep_la3.txt 2>&1 <<'S'
l_a8_rb8_la3_a||--model qwen7b-fp8a8 --wide-row-blocks 8
l_a8_rb8_la3_b||--model qwen7b-fp8a8 --wide-row-blocks 8
S
cat gpurun_out/r04l_sweep_la3.txt
bash tools/profile_bench.sh r04l_fp8a8_rb8_la3_1lane --model qwen7b-fp8a8 --lanes 1 --cohort 4 --wide-row-blocks 8
python tools/stats_summary.py gpurun_out/kernel_stats_r04l_fp8a8_rb8_la3_1lane.csv 8
cp /tmp/lib_la2.so vispec_amd/libvispec_hip.so
