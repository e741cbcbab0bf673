export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd /tmp
mkdir -p $R/gpurun_out/r01b
python3 $R/bench.py --layer-report $R/gpurun_out/r01b/per_launch.csv > $R/gpurun_out/r01b/bench_line.json 2> $R/gpurun_out/r01b/bench.err
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r01b/serial -o serial -- python3 $R/bench.py --serial --no-cpu-baseline > $R/gpurun_out/r01b/serial_bench_line.json 2> $R/gpurun_out/r01b/serial.err
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r01b/default -o default -- python3 $R/bench.py --no-cpu-baseline > $R/gpurun_out/r01b/default_bench_line.json 2> $R/gpurun_out/r01b/default.err
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/r01b/fetch -o fetch -- python3 $R/bench.py --serial --steps 1 --warmup 1 --no-cpu-baseline > /dev/null 2> $R/gpurun_out/r01b/fetch.err
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/r01b/write -o write -- python3 $R/bench.py --serial --steps 1 --warmup 1 --no-cpu-baseline > /dev/null 2> $R/gpurun_out/r01b/write.err
cd $R
find gpurun_out/r01b -name "*.csv" | head -30
F=$(find gpurun_out/r01b/fetch -name "*counter_collection.csv" | head -1); W=$(find gpurun_out/r01b/write -name "*counter_collection.csv" | head -1)
python3 tools/pmc_summary.py $F $W hands_light gpurun_out/r01b/pmc_hands_light.json
# keep only the small summaries
find gpurun_out/r01b -name "*kernel_trace.csv" -delete; find gpurun_out/r01b -name "*counter_collection.csv" -delete
du -sh gpurun_out/r01b
