#!/bin/bash
# usage (GPU box): bash tools/r3_build_ab.sh <tag> <sections> "<flagsA>" "<flagsB>" ...  -> one bench line per hipcc flag set
tag=$1; sec=$2; shift 2
i=0
for fl in "$@"; do
  i=$((i+1))
  DL_CXXFLAGS="$fl" python3 -m disenlink_amd.build --force > /dev/null 2> gpurun_out/${tag}_build$i.err || { tail -3 gpurun_out/${tag}_build$i.err; exit 1; }
  bash tools/r3_var.sh ${tag}_$i "DL_X=1" $sec | sed "s|DL_X=1|$fl|"
done
