# usage (GPU box): bash tools/ab_bench.sh <workload> <bz> <variant> [<variant> ...]   ("-" = the in-tree library)
# alternates the variants twice on ONE box (boxes differ by ~6 %): value / serial ms / roofline frac per run
R=$GRAFT_REPO_ROOT; WL=$1; BZ=$2; shift 2
for rep in 1 2; do for v in "$@"; do
  if [ "$v" = "-" ]; then unset HANDS_HIP_LIB; else export HANDS_HIP_LIB=$R/build_ab/$v.so; fi
  python3 $R/bench.py --workload $WL --bz $BZ --no-cpu-baseline --no-also --no-pmc --steps ${STEPS:-10} --warmup 3 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$v', 'rep$rep', d['value'], 'ms', d['ms_per_step'], 'serial_ms', d['serial']['ms_per_step'], 'frac', d['roofline']['frac'], 'kernel_ms', d['roofline']['kernel_ms_per_step'])"
done; done
