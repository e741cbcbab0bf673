# usage (GPU box): bash tools/experiments/ab_hon_depth.sh    -- handoccnet_light bz 32, forwards in flight (pipeline_depth), alternated twice on ONE box
R=$GRAFT_REPO_ROOT
for rep in 1 2; do for c in 3 2 4 5; do
  HANDS_PIPE_DEPTH=$c python3 $R/bench.py --workload handoccnet_light --bz 32 --no-cpu-baseline --no-also --no-pmc --steps 20 --warmup 5 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('depth $c', 'rep$rep', d['value'], 'ms', d['ms_per_step'])"
done; done
