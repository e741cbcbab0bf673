# usage (GPU box): bash tools/ab_conv.sh <variant> [<variant> ...]   ("-" = the in-tree library)
# the four 3x3 / stride-1 trunk shapes of hands_light through tools/bench_conv.py, variants alternated twice on ONE box
R=$GRAFT_REPO_ROOT
export HANDS_BENCH_SHAPES="${SHAPES:-512,256,14,256,3,1,1,0;512,128,28,128,3,1,1,0;512,64,56,64,3,1,1,0;256,512,7,512,3,1,1,0}"
for rep in 1 2; do for v in "$@"; do
  if [ "$v" = "-" ]; then unset HANDS_HIP_LIB; else export HANDS_HIP_LIB=$R/build_ab/$v.so; fi
  echo "== $v rep$rep"; python3 $R/tools/bench_conv.py ${REPS:-20} 2>/dev/null | awk '{print "   ", $0}'
done; done
