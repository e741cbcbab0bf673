# usage (GPU box): bash tools/experiments/w4_band_pmc.sh   -> FETCH_SIZE / WRITE_SIZE per conv_wino4 launch, band 1 against the default
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; export HANDS_HIP_LIB=$R/build_ab/w4env.so   # built with -DHANDS_W4_BAND_ENV
for CTR in FETCH_SIZE WRITE_SIZE; do
  O=/tmp/w4band_$CTR; rm -rf $O; mkdir -p $O; cd /tmp
  rocprofv3 --pmc $CTR --output-format csv -d $O -o f -- python3 $R/tools/experiments/w4_band.py --sequence > /dev/null 2> $O/err
  F=$(find $O -name "*counter_collection.csv" | head -1)
  python3 - "$F" "$CTR" <<'PY'
import csv,sys
rows=[(int(r["Dispatch_Id"]), float(r["Counter_Value"])) for r in csv.DictReader(open(sys.argv[1])) if "conv_wino4" in r["Kernel_Name"] and r["Counter_Name"]==sys.argv[2]]
rows.sort()
names=["layer1 64ch 56x56","layer2 128ch 28x28","layer3 256ch 14x14","layer4 512ch 7x7"]
for i,(d,v) in enumerate(rows):
    print(sys.argv[2], names[i//2], "band 1 (round 5)" if i%2==0 else "banded (round 6)", "KB", round(v,1), "-> GB (x2 for FETCH on gfx950)", round((2 if sys.argv[2]=="FETCH_SIZE" else 1)*v*1024/1e9,4))
PY
  cd $R
done
