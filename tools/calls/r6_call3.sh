# round 6, call 3: XCD partition of the lanes (CU-masked streams): probe of the cohort GEMMs on 4 streams plain vs partitioned (with the
# XCC ids the workgroups report), then the bench line plain vs --xcd-partition, same box; the remaining c8 per-element bar; world-8 dry run
mkdir -p gpurun_out
VISPEC_LIB_VARIANT=wgclk timeout 600 python tools/cu_mask_probe.py 4 > gpurun_out/r06_cu_mask_probe_wgclk.txt 2>&1; tail -8 gpurun_out/r06_cu_mask_probe_wgclk.txt
timeout 600 python tools/cu_mask_probe.py 4 > gpurun_out/r06_cu_mask_probe.txt 2>&1; tail -6 gpurun_out/r06_cu_mask_probe.txt
timeout 600 python tools/cu_mask_probe.py 2 > gpurun_out/r06_cu_mask_probe_2lanes.txt 2>&1; tail -6 gpurun_out/r06_cu_mask_probe_2lanes.txt
bash tools/sweep.sh > gpurun_out/r06a_sweep.txt 2>&1 <<'S'
a_plain||
a_xcd||--xcd-partition
a_plain_nv||--no-vision-in-loop
a_xcd_nv||--xcd-partition --no-vision-in-loop
a_xcd_nv_b||--xcd-partition --no-vision-in-loop
a_plain_nv_b||--no-vision-in-loop
S
cat gpurun_out/r06a_sweep.txt
timeout 900 python -m pytest tests/test_c8_gpu.py -q -m gpu -k "fp64_product" --tb=line 2>&1 | tail -5
timeout 1700 python -m pytest tests/test_world8_gpu.py -x -q -m gpu -s 2>&1 | tail -15 > gpurun_out/r06_world8_test.txt; cat gpurun_out/r06_world8_test.txt
