# usage (GPU box): bash tools/ab_link.sh <variant|-> ...  -> the fused seam launches' time per hands_light forward (serial pass)
R=$GRAFT_REPO_ROOT
for v in "$@"; do
  if [ "$v" = "-" ]; then unset HANDS_HIP_LIB; else export HANDS_HIP_LIB=$R/build_ab/$v.so; fi
  python3 $R/bench.py --no-cpu-baseline --no-also --steps 5 --warmup 2 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['roofline']['kernels']
print('$v', d['value'], 'serial', d['serial']['ms_per_step'], 'frac', d['roofline']['frac'], k.get('bottleneck_link_kernel'))"
done
