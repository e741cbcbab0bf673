# usage (GPU box): bash tools/experiments/ab_trunk_chunks.sh    -- hands_light bz 256, trunk_chunks (global, hand) settings alternated twice on ONE box
R=$GRAFT_REPO_ROOT
for rep in 1 2; do for c in 1,2 1,1 2,2 1,3 1,4 2,4; do
  HANDS_CHUNKS=$c python3 $R/bench.py --workload hands_light --bz 256 --no-cpu-baseline --no-also --no-pmc --steps 10 --warmup 3 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('chunks $c', 'rep$rep', d['value'], 'ms', d['ms_per_step'])"
done; done
