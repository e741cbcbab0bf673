# usage (on the GPU box): bash tools/pmc_ta.sh <tag> <workload> [bz]   -- texture-unit busy fraction per kernel family
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$1; mkdir -p $O; cd /tmp
timeout 300 rocprofv3 --pmc TA_TA_BUSY_sum GRBM_GUI_ACTIVE --output-format csv -d $O/ta -o p -- python3 $R/bench.py --workload $2 --bz ${3:-256} --no-cpu-baseline --no-also --no-sweep --serial --steps 1 --warmup 1 > /dev/null 2> $O/ta.err
echo "rc $?"
cd $R
python3 - $O <<'PY'
import csv, glob, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for f in glob.glob(sys.argv[1] + "/ta/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        fam = next((x for x in ("conv_wino", "conv_igemm_sk", "conv_igemm_splitk", "conv_igemm", "stem_pool", "flash_attention64", "attention_kernel", "mano_heads") if x in k), "other")
        acc[fam][r["Counter_Name"]] += float(r["Counter_Value"]); n[fam] += 1
for fam, c in sorted(acc.items()):
    if c["GRBM_GUI_ACTIVE"] > 0:
        print(f"{fam}: {n[fam] // 2} dispatches, texture units busy {100 * (c['TA_TA_BUSY_sum'] / 256) / (c['GRBM_GUI_ACTIVE'] / 8):.1f} % of GPU-active cycles")
PY
