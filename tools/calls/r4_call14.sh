# round 4, call 14: W8A8 with the quantisation of the q|k|v and gate|up inputs fused into the split-K reduce + norm (two launches per layer less),
# eight row blocks by default; full-size tests of the W8A8 model; same-box A/B against W8A16; per-kernel times, one lane and four
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_fp8a8_gpu.py -q > gpurun_out/r04m_pytest_fp8a8.txt 2>&1; tail -6 gpurun_out/r04m_pytest_fp8a8.txt | cut -c1-250
timeout 1500 python -m pytest tests/test_full_size_gpu.py -q -k "fp8a8" -s > gpurun_out/r04m_pytest_fullsize_fp8a8.txt 2>&1; grep -a "full-width\|passed\|failed\|Error" gpurun_out/r04m_pytest_fullsize_fp8a8.txt | cut -c1-400 | tail -12
bash tools/sweep.sh > gpurun_out/r04m_sweep.txt 2>&1 <<'S'
m_fp8_a||--model qwen7b-fp8
m_a8_a||--model qwen7b-fp8a8
m_fp8_b||--model qwen7b-fp8
m_a8_b||--model qwen7b-fp8a8
m_a8_rb4||--model qwen7b-fp8a8 --wide-row-blocks 4
S
cat gpurun_out/r04m_sweep.txt
bash tools/profile_bench.sh r04m_fp8a8_1lane --model qwen7b-fp8a8 --lanes 1 --cohort 4 --wide-row-blocks 84
python tools/stats_summary.py gpurun_out/kernel_stats_r04m_fp8a8_1lane.csv 12
bash tools/profile_bench.sh r04m_fp8a8_4lanes --model qwen7b-fp8a8
python tools/stats_summary.py gpurun_out/kernel_stats_r04m_fp8a8_4lanes.csv 12
