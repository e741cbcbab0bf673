# usage (on the GPU box, from the repo root): bash tools/profile_round.sh <tag> <workload> [bz]
# rocprofv3 kernel stats (serial + shipped mode) and PMC passes (FETCH_SIZE, WRITE_SIZE, MFMA busy, L2 hit/miss) of one
# workload at one batch size; keeps only the small summaries under gpurun_out/<tag>/<workload>_bz<bz>/ (copy what is to
# be judged into profiles/).  PMC passes are separate runs without --kernel-trace (gpurun refuses the combination).
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
TAG=$1; WL=$2; BZ=${3:-0}
if [ "$BZ" = "0" ]; then case $WL in hamer_light) BZ=64;; handoccnet_light) BZ=256;; mano_lbs) BZ=1024;; *) BZ=256;; esac; fi
O=$R/gpurun_out/$TAG/${WL}_bz$BZ
mkdir -p $O
cd /tmp
COMMON="--workload $WL --bz $BZ --no-cpu-baseline --no-also --no-pmc"   # (the first, unprofiled run below takes the same-run traffic passes)
python3 $R/bench.py --workload $WL --bz $BZ --no-also --layer-report $O/per_launch.csv > $O/bench_line.json 2> $O/bench.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/serial -o serial -- python3 $R/bench.py $COMMON --serial --steps 5 --warmup 2 > $O/serial_bench_line.json 2> $O/serial.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/default -o default -- python3 $R/bench.py $COMMON --steps 5 --warmup 2 > $O/default_bench_line.json 2> $O/default.err
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/fetch -o fetch -- python3 $R/bench.py $COMMON --serial --steps 1 --warmup 1 > /dev/null 2> $O/fetch.err
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/write -o write -- python3 $R/bench.py $COMMON --serial --steps 1 --warmup 1 > /dev/null 2> $O/write.err
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $O/mfma -o mfma -- python3 $R/bench.py $COMMON --serial --steps 1 --warmup 1 > /dev/null 2> $O/mfma.err
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $O/l2 -o l2 -- python3 $R/bench.py $COMMON --serial --steps 1 --warmup 1 > /dev/null 2> $O/l2.err
if [ "$WL" != "mano_lbs" ]; then
# the same two counters over the SHIPPED mode's launches (multi-stream: plain kernels, no stream-K; rocprofv3 serialises the
# dispatches of a counter pass, so per-dispatch values stay meaningful): HANDS_BENCH_SHIPPED_ONLY keeps the one-stream passes
# of bench.py out of the process
HANDS_BENCH_SHIPPED_ONLY=1 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/fetch_s -o fetch -- python3 $R/bench.py $COMMON --steps 1 --warmup 1 > /dev/null 2> $O/fetch_s.err
HANDS_BENCH_SHIPPED_ONLY=1 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/write_s -o write -- python3 $R/bench.py $COMMON --steps 1 --warmup 1 > /dev/null 2> $O/write_s.err
fi
cd $R
if [ "$WL" != "mano_lbs" ]; then
F=$(find $O/fetch_s -name "*counter_collection.csv" | head -1); W=$(find $O/write_s -name "*counter_collection.csv" | head -1)
python3 tools/pmc_summary.py $F $W $WL $O/pmc_${WL}_bz${BZ}_shipped.json $BZ shipped
F=$(find $O/fetch -name "*counter_collection.csv" | head -1); W=$(find $O/write -name "*counter_collection.csv" | head -1)
python3 tools/pmc_summary.py $F $W $WL $O/pmc_${WL}_bz$BZ.json $BZ
fi
M=$(find $O/mfma -name "*counter_collection.csv" | head -1)
python3 tools/pmc_mfma.py $M > $O/pmc_mfma_busy.txt; cat $O/pmc_mfma_busy.txt
M=$(find $O/l2 -name "*counter_collection.csv" | head -1)
python3 tools/pmc_l2.py $M > $O/pmc_l2_hit.txt; cat $O/pmc_l2_hit.txt
for d in serial default; do S=$(find $O/$d -name "*kernel_stats.csv" | head -1); cp $S $O/${d}_kernel_stats.csv; done
rm -rf $O/serial $O/default $O/fetch $O/write $O/fetch_s $O/write_s $O/mfma $O/l2
du -sh $O
