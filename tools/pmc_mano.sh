# usage (on the GPU box): bash tools/pmc_mano.sh <tag> [bz list, default 4096] [launches, default 3]
# one rocprofv3 --pmc pass per counter pair over tools/bench_mano_kernel.py, mano_heads_kernel only.  The texture-unit counters
# are slow to collect (minutes per pass even for a handful of dispatches): each pass runs under its own timeout.
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$1; mkdir -p $O; cd /tmp
i=0
for SET in "TA_TA_BUSY_sum GRBM_GUI_ACTIVE" "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum" "TCP_PENDING_STALL_CYCLES_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum"; do
  i=$((i+1))
  timeout ${4:-150} rocprofv3 --pmc $SET --output-format csv -d $O/p$i -o p -- python3 $R/tools/bench_mano_kernel.py ${2:-4096} ${3:-3} > /dev/null 2> $O/p$i.err
  echo "pass $i ($SET): rc $?"
done
cd $R
python3 - $O <<'PY'
import csv, glob, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(sys.argv[1] + "/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "mano_heads" in r["Kernel_Name"]:
            acc[r["Grid_Size"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for g in sorted(acc, key=int):
    print("grid", g, {k: round(sum(v) / len(v)) for k, v in sorted(acc[g].items())})
PY
