# round 5, call 3: kernel stats of the cohort-8 line (4 lanes and 1 lane), split factor / draft form A-Bs
mkdir -p gpurun_out
bash tools/profile_bench.sh r05_4lanes_cohort8 --lanes 4 --cohort 8
python tools/stats_summary.py gpurun_out/kernel_stats_r05_4lanes_cohort8.csv 32
bash tools/profile_bench.sh r05_1lane_cohort8 --lanes 1 --cohort 8
python tools/stats_summary.py gpurun_out/kernel_stats_r05_1lane_cohort8.csv 32
bash tools/sweep.sh > gpurun_out/r05c_sweep.txt 2>&1 <<'S'
c_l4c8||--lanes 4 --cohort 8
c_l4c8_s8|VISPEC_C8_SPLIT=8|--lanes 4 --cohort 8
c_l4c8_s2|VISPEC_C8_SPLIT=2|--lanes 4 --cohort 8
c_l4c8_noslab|VISPEC_DRAFT_SLAB=0|--lanes 4 --cohort 8
c_l5c8||--lanes 5 --cohort 8
c_l4c6||--lanes 4 --cohort 6
S
cat gpurun_out/r05c_sweep.txt
