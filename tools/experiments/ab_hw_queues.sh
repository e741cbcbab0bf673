# HIP maps streams onto GPU_MAX_HW_QUEUES hardware queues (default 4): do more queues help the multi-stream forwards?  (GPU box)
for i in 1 2; do for q in 4 8 16; do
GPU_MAX_HW_QUEUES=$q python bench.py --no-also --no-pmc --no-cpu-baseline --steps 12 --warmup 4 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('hands_light       queues $q', d['value'], d['ms_per_step'])"
GPU_MAX_HW_QUEUES=$q python bench.py --workload handoccnet_light --bz 32 --no-also --no-pmc --no-cpu-baseline --steps 40 --warmup 8 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('handoccnet_light  queues $q', d['value'], d['ms_per_step'])"
done; done
