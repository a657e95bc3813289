# round 5, call 21: three groups of lookahead in the cohort-8 kernel (12 KiB of W per wave + 96 KiB of X in flight) vs two, same box
mkdir -p gpurun_out
VISPEC_LIB_VARIANT=la3 timeout 900 python -m pytest tests/test_c8_gpu.py -x -q -m gpu -k "fp64 or depend" 2>&1 | tail -2
VISPEC_LIB_VARIANT=la3 timeout 600 python tools/c8_bench.py 2>&1 | grep -v "^check" | tail -5
bash tools/sweep.sh > gpurun_out/r05m_sweep.txt 2>&1 <<'S'
m_la2||--no-vision-in-loop
m_la3|VISPEC_LIB_VARIANT=la3|--no-vision-in-loop
m_la2_b||--no-vision-in-loop
m_la3_b|VISPEC_LIB_VARIANT=la3|--no-vision-in-loop
S
cat gpurun_out/r05m_sweep.txt
