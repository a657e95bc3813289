mkdir -p gpurun_out
bash tools/sweep.sh <<'S'
base||
kpw1024|VISPEC_ATT_KPW=1024|
kpw768|VISPEC_ATT_KPW=768|
kpw384|VISPEC_ATT_KPW=384|
hwq16|GPU_MAX_HW_QUEUES=16|
hwq4|GPU_MAX_HW_QUEUES=4|
base2||
S
