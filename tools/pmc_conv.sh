# usage: bash tools/pmc_conv.sh <tag> <fp32|bf16x3>   SQ counters of the conv micro-benchmark (first shape only)
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; TAG=$1; M=$2; O=$R/gpurun_out/$TAG
mkdir -p $O; cd /tmp
HANDS_BENCH_ONE=1 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $O/a -o a -- python3 $R/tools/bench_conv.py 3 $M > /dev/null 2> $O/a.err
HANDS_BENCH_ONE=1 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS --output-format csv -d $O/b -o b -- python3 $R/tools/bench_conv.py 3 $M > /dev/null 2> $O/b.err
cd $R
python3 - <<PY
import csv,glob,collections
for d in ("a","b"):
    for fn in glob.glob("$O/%s/**/*counter_collection.csv"%d, recursive=True):
        acc=collections.defaultdict(float); n=set()
        for r in csv.DictReader(open(fn)):
            if "conv_igemm" in r["Kernel_Name"] or "conv_wino" in r["Kernel_Name"]:
                acc[r["Counter_Name"]]+=float(r["Counter_Value"]); n.add(r["Dispatch_Id"])
        print("$M", d, len(n), {k: "%.4g"%(v/len(n)) for k,v in acc.items()})
PY
rm -rf $O/a $O/b
