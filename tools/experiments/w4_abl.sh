# usage (GPU box): bash tools/experiments/w4_abl.sh [images]   timing-only ablations of conv_wino4 (build_ab/w4abl<n>.so, tools/build_variant.sh with EXTRA_FLAGS=-DW4_ABL=n)
R=$GRAFT_REPO_ROOT
for v in - w4abl1 w4abl2 w4abl3 w4abl4 w4abl6; do
  if [ "$v" = "-" ]; then unset HANDS_HIP_LIB; else export HANDS_HIP_LIB=$R/build_ab/$v.so; fi
  echo "== $v"; python3 $R/tools/wino4_check.py --time-only --images ${1:-512} 2>&1 | tail -4
done
