# round 6, call 5: do lanes that stream the SAME weights in lockstep run faster (Infinity Cache hits instead of HBM)?  + the full-size
# checkpoint test (published CLIP key names under transformers 5.x) + the member-slot error test
mkdir -p gpurun_out
timeout 900 python tools/cu_mask_probe.py 4 > gpurun_out/r06_shared_weights_probe_4lanes.txt 2>&1; tail -5 gpurun_out/r06_shared_weights_probe_4lanes.txt
timeout 900 python tools/cu_mask_probe.py 2 > gpurun_out/r06_shared_weights_probe_2lanes.txt 2>&1; tail -5 gpurun_out/r06_shared_weights_probe_2lanes.txt
timeout 600 python -m pytest tests/test_wide_gpu.py -x -q -m gpu -k "member_slots" --tb=short 2>&1 | tail -5
timeout 2400 python -m pytest tests/test_checkpoint_full_size_gpu.py -x -q -m gpu -s --tb=short 2>&1 | tail -25 > gpurun_out/r06_checkpoint_full_size.txt; cat gpurun_out/r06_checkpoint_full_size.txt
