R=$GRAFT_REPO_ROOT
export HANDS_BENCH_SHAPES="512,128,28,512,1,1,0,1;512,64,56,256,1,1,0,1;512,256,56,64,1,1,0,0;512,512,28,128,1,1,0,0;512,256,14,1024,1,1,0,1;512,1024,14,256,1,1,0,0;512,256,56,128,1,1,0,0"
for v in - ptocc3; do for P in 0 1; do
  if [ "$v" = "-" ]; then unset HANDS_HIP_LIB; else export HANDS_HIP_LIB=$R/build_ab/$v.so; fi
  echo "== lib $v persistent $P"; HANDS_PERSISTENT=$P python3 $R/tools/bench_conv.py 20 2>/dev/null | awk '{print "   ", $0}'
done; done
