mkdir -p gpurun_out
bash tools/sweep.sh <<'S'
l4rb4||--lanes 4 --cohort 4
l3rb4||--lanes 3 --cohort 4
l5rb4||--lanes 5 --cohort 4
l3rb0||--lanes 3 --cohort 4 --wide-row-blocks 0
l4rb0||--lanes 4 --cohort 4 --wide-row-blocks 0
l2rb0||--lanes 2 --cohort 4 --wide-row-blocks 0
l6rb4||--lanes 6 --cohort 4
l4rb4b||--lanes 4 --cohort 4
S
