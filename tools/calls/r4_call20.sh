# round 4, call 20: can the small dependent kernels share a CU with an eight-row-block GEMM workgroup (96 free VGPRs per SIMD)?
# split-K reduce with 256 threads (72 VGPRs x 1 wave per SIMD fits; 512 threads = 2 waves per SIMD do not), attention merge capped at 96 VGPRs
mkdir -p gpurun_out
bash tools/sweep.sh > gpurun_out/r04s_sweep.txt 2>&1 <<'S'
s_base_a||
s_red256_a|VISPEC_REDUCE_THREADS=256|
s_base_b||
s_red256_b|VISPEC_REDUCE_THREADS=256|
S
cp vispec_amd/libvispec_hip.so /tmp/lib_base.so; cp vispec_amd/libvispec_hip_v.so vispec_amd/libvispec_hip.so
bash tools/sweep.sh >> gpurun_out/r04s_sweep.txt 2>&1 <<'S'
s_attn96_a||
s_attn96_red256_a|VISPEC_REDUCE_THREADS=256|
s_attn96_b||
s_attn96_red256_b|VISPEC_REDUCE_THREADS=256|
S
cp /tmp/lib_base.so vispec_amd/libvispec_hip.so
cat gpurun_out/r04s_sweep.txt
