# usage (GPU box): bash tools/fetch_ab.sh <workload> <bz> <variant|-> ...   -> FETCH_SIZE-derived read GB per conv_igemm launch per variant
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; WL=$1; BZ=$2; shift 2
for v in "$@"; do
  if [ "$v" = "-" ]; then unset HANDS_HIP_LIB; else export HANDS_HIP_LIB=$R/build_ab/$v.so; fi
  O=/tmp/fetch_$v; rm -rf $O; mkdir -p $O; cd /tmp
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O -o f -- python3 $R/bench.py --workload $WL --bz $BZ --no-cpu-baseline --no-also --serial --steps 1 --warmup 1 > /dev/null 2> $O/err
  F=$(find $O -name "*counter_collection.csv" | head -1)
  python3 - "$F" "$v" <<'PY'
import csv,sys
tot=0.0; disp=set()
for r in csv.DictReader(open(sys.argv[1])):
    if "conv_igemm" in r["Kernel_Name"] and r["Counter_Name"]=="FETCH_SIZE":
        tot+=float(r["Counter_Value"]); disp.add(r["Dispatch_Id"])
print(sys.argv[2], "conv_igemm dispatches", len(disp), "read GB per launch (2 x FETCH_SIZE)", round(2*tot*1024/len(disp)/1e9,4))
PY
  cd $R
done
