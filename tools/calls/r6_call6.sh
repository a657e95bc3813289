# round 6, call 6: shared-activation probe (L2 footprint of the lanes' X blocks) + member-slot test
mkdir -p gpurun_out
timeout 900 python tools/cu_mask_probe.py 4 > gpurun_out/r06_shared_x_probe_4lanes.txt 2>&1; tail -5 gpurun_out/r06_shared_x_probe_4lanes.txt
timeout 900 python tools/cu_mask_probe.py 4 > gpurun_out/r06_shared_x_probe_4lanes_b.txt 2>&1; tail -5 gpurun_out/r06_shared_x_probe_4lanes_b.txt
timeout 600 python -m pytest tests/test_wide_gpu.py -x -q -m gpu -k "member_slots" --tb=short 2>&1 | tail -5
