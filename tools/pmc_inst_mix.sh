# usage (GPU box): bash tools/pmc_inst_mix.sh <workload> <bz>   -> instruction mix per kernel family of two one-stream forwards
# (rocprofv3 --pmc, no trace domains): VALU / MFMA / LDS / SALU / VMEM instructions per dispatch, VALU-active and MFMA-busy cycles.
# fp32 MFMA and VALU share a SIMD's one fp32 pipe on gfx950 (SQ_VALU_MFMA_COEXEC_CYCLES = 0): VALU instructions per MFMA is what a
# kernel pays on top of its matrix time.
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; WL=$1; BZ=$2
O=/tmp/imix; rm -rf $O; mkdir -p $O; cd /tmp
HANDS_BENCH_PMC_CHILD=1 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_COEXEC_CYCLES --output-format csv -d $O/a -o a -- python3 $R/bench.py --pmc-child --workload $WL --bz $BZ > /dev/null 2> $O/a.err
HANDS_BENCH_PMC_CHILD=1 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY --output-format csv -d $O/b -o b -- python3 $R/bench.py --pmc-child --workload $WL --bz $BZ > /dev/null 2> $O/b.err
cd $R
python3 - "$WL" "$BZ" <<'PY'
import csv,glob,collections,sys,re
fam=lambda n: next((f for f in ("conv_igemm_group","conv_igemm_sk","conv_igemm","conv_wino4","conv_wino","stem_pool","flash_attention64","attention_kernel","mano_heads","layernorm") if f in n), None)
acc=collections.defaultdict(lambda: collections.defaultdict(float)); cnt=collections.defaultdict(set)
for d in ("a","b"):
    for fn in glob.glob(f"/tmp/imix/{d}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(fn)):
            f=fam(r["Kernel_Name"])
            if f: acc[f][r["Counter_Name"]]+=float(r["Counter_Value"]); cnt[f].add(r["Dispatch_Id"]) if d=="a" else None
print(f"{sys.argv[1]} bz {sys.argv[2]}: per kernel family (sums over two one-stream forwards)")
for f,c in acc.items():
    mf=max(c["SQ_INSTS_MFMA"],1)
    busy=(c["SQ_VALU_MFMA_BUSY_CYCLES"]/1024)/max(c["GRBM_GUI_ACTIVE"]/8,1)
    print(f"  {f:18s} dispatches {len(cnt[f]):4d}  per MFMA: VALU {c['SQ_INSTS_VALU']/mf:5.2f}  LDS {c['SQ_INSTS_LDS']/mf:5.2f}  SALU {c['SQ_INSTS_SALU']/mf:5.2f}  VMEM {(c['SQ_INSTS_VMEM_RD']+c['SQ_INSTS_VMEM_WR'])/mf:5.2f} | VALU-active cycles / MFMA-busy cycles {c['SQ_ACTIVE_INST_VALU']/max(c['SQ_VALU_MFMA_BUSY_CYCLES'],1):.3f}  MFMA-busy {100*busy:.1f} %  coexec {c['SQ_VALU_MFMA_COEXEC_CYCLES']:.0f}  wait/wave-cycles {c['SQ_WAIT_INST_ANY']/max(c['SQ_WAVE_CYCLES'],1):.2f}")
PY
