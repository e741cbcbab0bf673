# usage (GPU box): [PMC_MODE=--serial] bash tools/pmc_positions.sh   -> gpurun_out/pos/pmc_pos.json: fabric traffic per launch POSITION of a
# hands_light forward (joins with per_launch.csv); default = the shipped multi-stream launches (plain kernels), --serial = the
# one-stream mode (stream-K launches where the library picks them)
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/pos; mkdir -p $O; cd /tmp
C="--workload hands_light --bz 256 --no-cpu-baseline --no-also"
python3 $R/bench.py $C --layer-report $O/per_launch.csv > $O/line.json 2>/dev/null
if [ "$PMC_MODE" != "--serial" ]; then export HANDS_BENCH_SHIPPED_ONLY=1; fi
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/fetch -o fetch -- python3 $R/bench.py $C $PMC_MODE --steps 1 --warmup 1 > /dev/null 2> $O/fetch.err
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/write -o write -- python3 $R/bench.py $C $PMC_MODE --steps 1 --warmup 1 > /dev/null 2> $O/write.err
cd $R
F=$(find $O/fetch -name "*counter_collection.csv" | head -1); W=$(find $O/write -name "*counter_collection.csv" | head -1)
python3 tools/pmc_summary.py $F $W hands_light $O/pmc_pos.json 256 | tail -1 | cut -c1-200
rm -rf $O/fetch $O/write
