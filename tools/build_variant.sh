#!/bin/bash
# usage: tools/build_variant.sh <name> [git-ref|-] [replacement file] [file name it replaces, default conv_igemm.hip]
#   -> build_ab/<name>.so from the working tree (ref "-" or empty) or from csrc/ at <git-ref>
# A/B on ONE box (boxes differ by ~6 %): HANDS_HIP_LIB=build_ab/<name>.so python bench.py ...
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
N=$1; REF=$2; REPL=$3
if [ "$REF" = "-" ]; then REF=""; fi
D=$R/build_ab/src_$N
rm -rf $D; mkdir -p $D
if [ -n "$REF" ]; then
  for f in $(git -C $R ls-tree --name-only $REF hands_amd/csrc/); do git -C $R show $REF:$f > $D/$(basename $f); done
  git -C $R show $REF:include/hands_hip.h > $D/hands_hip.h
else
  cp $R/hands_amd/csrc/*.hip $R/hands_amd/csrc/*.h $R/hands_amd/csrc/*.cpp $D/; cp $R/include/hands_hip.h $D/
fi
if [ -n "$REPL" ]; then cp $REPL $D/${4:-conv_igemm.hip}; fi
cd $D
# the source hash the library reports (hands_csrc_sha16): this variant's own sources + flags, never the tree's
printf '#define HANDS_CSRC_SHA16 "%s"\n' "$( (cat $(ls *.hip *.h *.cpp | sort); echo "$EXTRA_FLAGS") | sha256sum | cut -c1-16)" > csrc_sha.inc
FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -I. -fno-fast-math -ffp-contract=off -Wno-unused-function $EXTRA_FLAGS"   # EXTRA_FLAGS: e.g. -DHANDS_EPI_SCALAR_ADDS
OBJS=""
for f in *.hip; do X=""; if [ "$f" = "conv_wino.hip" ]; then X="-fno-slp-vectorize"; fi; /opt/rocm/bin/hipcc $FLAGS $X -c $f -o ${f%.hip}.o & OBJS="$OBJS ${f%.hip}.o"; done
wait
if [ -f pack.cpp ]; then /opt/rocm/bin/hipcc -O2 -std=c++17 -fPIC -I. -fno-fast-math -ffp-contract=off -x c++ -c pack.cpp -o pack.o; OBJS="$OBJS pack.o"; fi
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $OBJS -o $R/build_ab/$N.so
rm -rf $D
ls -la $R/build_ab/$N.so
