# usage: bash tools/pmc_shape.sh <tag> "<B,Cin,H,Cout,k,stride,pad,res>" [...]   SQ / TCP / TCC counters of ONE conv shape per run
# (separate --pmc passes, no trace domains); prints per-dispatch averages over the conv_igemm launches
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; TAG=$1; shift; O=$R/gpurun_out/$TAG
mkdir -p $O
SETS=(
"SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_COEXEC_CYCLES"
"SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS"
"SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_IFETCH"
"TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TA_BUSY_avr"
"TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_TAG_STALL_sum"
)
for SH in "$@"; do
  export HANDS_BENCH_SHAPES="$SH"
  i=0
  for S in "${SETS[@]}"; do
    cd /tmp
    rocprofv3 --pmc $S --output-format csv -d $O/p$i -o p -- python3 $R/tools/bench_conv.py 3 > /dev/null 2> $O/p$i.err
    i=$((i+1))
  done
  cd $R
  python3 - <<PY
import csv,glob,collections
acc=collections.defaultdict(float); cnt=collections.defaultdict(set)
for fn in glob.glob("$O/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(fn)):
        if "conv_igemm" in r["Kernel_Name"] or "conv_wino" in r["Kernel_Name"]:
            acc[r["Counter_Name"]]+=float(r["Counter_Value"]); cnt[r["Counter_Name"]].add(r["Dispatch_Id"])
print("shape $SH")
for k in sorted(acc): print("  %-34s %.5g" % (k, acc[k]/max(len(cnt[k]),1)))
PY
  rm -rf $O/p*
done
