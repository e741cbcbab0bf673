# usage: bash tools/pmc_only.sh <tag> <workload>   (FETCH_SIZE / WRITE_SIZE passes only; honours HANDS_HIP_LIB)
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; TAG=$1; WL=$2; O=$R/gpurun_out/$TAG/$WL
mkdir -p $O; cd /tmp
COMMON="--workload $WL --bz ${3:-256} --no-cpu-baseline --no-also --serial --steps 1 --warmup 1"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/fetch -o fetch -- python3 $R/bench.py $COMMON > /dev/null 2> $O/fetch.err
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/write -o write -- python3 $R/bench.py $COMMON > /dev/null 2> $O/write.err
cd $R
F=$(find $O/fetch -name "*counter_collection.csv" | head -1); W=$(find $O/write -name "*counter_collection.csv" | head -1)
python3 tools/pmc_summary.py $F $W $WL $O/pmc_$WL.json ${3:-256}
rm -rf $O/fetch $O/write
