# round 5, call 16: K / V rows of the tree attention loaded non-temporal (A/B, same box)
mkdir -p gpurun_out
bash tools/sweep.sh > gpurun_out/r05l_sweep.txt 2>&1 <<'S'
l_plain||--no-vision-in-loop
l_kvnt|VISPEC_LIB_VARIANT=kvnt|--no-vision-in-loop
l_plain_b||--no-vision-in-loop
l_kvnt_b|VISPEC_LIB_VARIANT=kvnt|--no-vision-in-loop
S
cat gpurun_out/r05l_sweep.txt
