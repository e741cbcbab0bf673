# usage (GPU box): bash tools/ab_conv1.sh <variant> ...   one pass per variant, compact output (us per shape)
R=$GRAFT_REPO_ROOT
export HANDS_BENCH_SHAPES="${SHAPES:-512,256,14,256,3,1,1,0;512,128,28,128,3,1,1,0;512,64,56,64,3,1,1,0;256,512,7,512,3,1,1,0}"
for v in "$@"; do
  if [ "$v" = "-" ]; then unset HANDS_HIP_LIB; else export HANDS_HIP_LIB=$R/build_ab/$v.so; fi
  printf "%-10s" $v; python3 $R/tools/bench_conv.py ${REPS:-20} 2>/dev/null | awk '/us/{printf "%8.1f us ", $11} /^sum/{printf " | sum %s ms\n", $2}'
done
