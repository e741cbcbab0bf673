# usage (GPU box): bash tools/launches_per_forward.sh <workload> <bz>
# Dispatches of ONE steady-state forward (one-stream mode), per kernel name: rocprofv3 --kernel-trace --stats over
# `bench.py --pmc-child` with 2 and with 4 forwards, (count4 - count2) / 2.  Weight uploads / packing (one blit
# `__amd_rocclr_copyBuffer` per tensor) and first-call allocations cancel out -- dividing a whole run's counts by the number of
# forwards does not do that (VERDICT r5 counted 48 / 69 copyBuffer "per forward" that way).
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; WL=$1; BZ=$2
for N in 2 4; do
  O=/tmp/lpf_$N; rm -rf $O; mkdir -p $O; cd /tmp
  HANDS_PMC_CHILD_FORWARDS=$N HANDS_BENCH_PMC_CHILD=1 rocprofv3 --kernel-trace --stats --output-format csv -d $O -o t -- python3 $R/bench.py --pmc-child --workload $WL --bz $BZ > /dev/null 2> $O/err
done
cd $R
python3 - "$WL" "$BZ" <<'PY'
import csv,glob,sys,collections
def counts(n):
    f=glob.glob(f"/tmp/lpf_{n}/**/*kernel_stats.csv", recursive=True)[0]
    return {r["Name"]: int(r["Calls"]) for r in csv.DictReader(open(f))}
c2,c4=counts(2),counts(4)
rows=sorted(((c4.get(k,0)-c2.get(k,0))/2, k) for k in set(c2)|set(c4))
tot=0; mfma=0; other=0
print(f"{sys.argv[1]} bz {sys.argv[2]}: dispatches per steady-state forward")
for n,k in reversed(rows):
    if n<=0: continue
    tot+=n
    is_mfma=any(t in k for t in ("conv_igemm","conv_wino","stem_pool","flash_attention","attention_kernel","mano_heads"))
    mfma+=n if is_mfma else 0; other+=0 if is_mfma else n
    print(f"  {n:7.1f}  {k[:110]}")
print(f"total {tot:.1f}  (MFMA kernels {mfma:.1f}, other {other:.1f}); one-time (2-forward run minus two forwards): copyBuffer", c2.get("__amd_rocclr_copyBuffer",0)-2*((c4.get("__amd_rocclr_copyBuffer",0)-c2.get("__amd_rocclr_copyBuffer",0))/2))
PY
