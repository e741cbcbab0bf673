# usage (GPU box): bash tools/pmc_link.sh [variant|-]   -> SQ / LDS / TCP counters of bottleneck_link_kernel alone
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; V=${1:--}
if [ "$V" != "-" ]; then export HANDS_HIP_LIB=$R/build_ab/$V.so; fi
cd /tmp
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" \
           "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS" \
           "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCC_HIT_sum TCC_MISS_sum"; do
  O=/tmp/pl; rm -rf $O
  rocprofv3 --pmc $set --output-format csv -d $O -o p -- python3 $R/tools/bench_link.py 512 ${C1:-64} 3 fused > /dev/null 2> /tmp/pl.err
  F=$(find $O -name "*counter_collection.csv" | head -1)
  python3 - "$F" <<'PY'
import csv,sys,collections
acc=collections.defaultdict(float); n=set()
for r in csv.DictReader(open(sys.argv[1])):
    if "bottleneck_link" in r["Kernel_Name"]:
        acc[r["Counter_Name"]]+=float(r["Counter_Value"]); n.add(r["Dispatch_Id"])
print({k: "%.4g" % (v/len(n)) for k,v in acc.items()}, "dispatches", len(n))
PY
done
