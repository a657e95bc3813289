# round 6, call 11: weight-stream lookahead of the cohort-8 kernel alone (-DC8_LW=3 / 4 at two ring groups): correctness of the variants, the
# kernel alone (tools/c8_bench.py), and the line, same box
mkdir -p gpurun_out
for v in lw3 lw4; do
  VISPEC_LIB_VARIANT=$v timeout 900 python -m pytest tests/test_c8_gpu.py -x -q -m gpu -k "fp64_product or do_not_depend or whole_loops" --tb=line 2>&1 | tail -3
done
timeout 900 python -m pytest tests/test_c8_gpu.py -x -q -m gpu -k "fp64_product or do_not_depend or whole_loops" --tb=line 2>&1 | tail -2
for v in "" lw3 lw4; do echo "== variant '$v'"; VISPEC_LIB_VARIANT=$v SHAPES=llava7b timeout 600 python tools/c8_bench.py 2>&1 | grep -v amdgpu.ids | tail -8; done > gpurun_out/r06_c8_lw_bench.txt 2>&1; cat gpurun_out/r06_c8_lw_bench.txt
bash tools/sweep.sh > gpurun_out/r06d_sweep.txt 2>&1 <<'S'
d_lw2||--no-vision-in-loop
d_lw3|VISPEC_LIB_VARIANT=lw3|--no-vision-in-loop
d_lw4|VISPEC_LIB_VARIANT=lw4|--no-vision-in-loop
d_lw2_b||--no-vision-in-loop
d_lw3_b|VISPEC_LIB_VARIANT=lw3|--no-vision-in-loop
d_lw4_b|VISPEC_LIB_VARIANT=lw4|--no-vision-in-loop
S
cat gpurun_out/r06d_sweep.txt
