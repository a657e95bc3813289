mkdir -p gpurun_out
bash tools/sweep.sh <<'S'
base||
kpw640|VISPEC_ATT_KPW=640|
kpw768|VISPEC_ATT_KPW=768|
kpw896|VISPEC_ATT_KPW=896|
base2||
kpw768b|VISPEC_ATT_KPW=768|
hwq16|GPU_MAX_HW_QUEUES=16|
S
