# round 5, call 25: read-once partials (split-K slabs, attention partials) loaded non-temporal by the kernels that finish them — A/B, same box
mkdir -p gpurun_out
bash tools/sweep.sh > gpurun_out/r05n_sweep.txt 2>&1 <<'S'
n_plain||--no-vision-in-loop
n_pnt|VISPEC_LIB_VARIANT=pnt|--no-vision-in-loop
n_plain_b||--no-vision-in-loop
n_pnt_b|VISPEC_LIB_VARIANT=pnt|--no-vision-in-loop
S
cat gpurun_out/r05n_sweep.txt
