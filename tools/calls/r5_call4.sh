# round 5, call 4: two-tile slab for the draft GEMMs of a cohort of 5..8: tests, then the line (slab vs tile-per-request draft), attention split sizes
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_c8_gpu.py tests/test_wide_gpu.py -x -q -m gpu 2>&1 | tail -5
bash tools/sweep.sh > gpurun_out/r05d_sweep.txt 2>&1 <<'S'
d_l4c8||--lanes 4 --cohort 8
d_l4c8_noslab|VISPEC_DRAFT_SLAB=0|--lanes 4 --cohort 8
d_l4c8_kpw1024|VISPEC_ATT_KPW=1024|--lanes 4 --cohort 8
d_l4c8_kpw256|VISPEC_ATT_KPW=256|--lanes 4 --cohort 8
d_l4c8_b||--lanes 4 --cohort 8
d_l3c8||--lanes 3 --cohort 8
S
cat gpurun_out/r05d_sweep.txt
