# usage (on the GPU box): bash tools/pmc_mano.sh <tag>    -- PMC passes over tools/bench_mano_kernel.py (mano_heads_kernel only)
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$1; mkdir -p $O; cd /tmp
i=0
for SET in "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_INST_ANY" "TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TCP_PENDING_STALL_CYCLES_sum" \
           "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F32 TD_TD_BUSY_sum"; do
  i=$((i+1))
  rocprofv3 --pmc $SET --output-format csv -d $O/p$i -o p -- python3 $R/tools/bench_mano_kernel.py ${2:-4096} ${3:-4} > /dev/null 2> $O/p$i.err
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
