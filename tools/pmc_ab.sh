# usage: bash tools/pmc_ab.sh "<shape>" <variant> [<variant> ...]   FETCH_SIZE of one conv shape under several library builds
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/pmcab; mkdir -p $O
export HANDS_BENCH_SHAPES="$1"; shift
for v in "$@"; do
  cd /tmp
  HANDS_HIP_LIB=$R/build_ab/$v.so rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/$v -o p -- python3 $R/tools/bench_conv.py 3 > /dev/null 2> $O/$v.err
  cd $R
  python3 - <<PY
import csv,glob,collections
acc=collections.defaultdict(float); cnt=collections.defaultdict(set)
for fn in glob.glob("$O/$v/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(fn)):
        if "conv_igemm" in r["Kernel_Name"]:
            k=r["Kernel_Name"][:60]+" grid="+r.get("Grid_Size","?")
            acc[k]+=float(r["Counter_Value"]); cnt[k].add(r["Dispatch_Id"])
for k in acc: print("$v", k, "FETCH GB/launch (x2 corrected): %.3f" % (2*acc[k]/len(cnt[k])/1e6))
PY
  rm -rf $O/$v
done
