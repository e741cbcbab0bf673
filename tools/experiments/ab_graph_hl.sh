# hands_light: eager multi-stream against hipGraph replay (depth 1), separate processes, alternating (GPU box)
for i in 1 2 3; do for g in 0 1; do
python bench.py --graph $g --no-also --no-pmc --no-cpu-baseline --steps 12 --warmup 4 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('graph $g', d['value'], d['ms_per_step'], d['config']['timed_mode'])"
done; done
