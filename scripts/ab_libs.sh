#!/bin/bash
# A/B of side builds (abl/*.so) over the conv sweep: scripts/ab_libs.sh "<lib list>" "<p0 tiles>" "<p1 tiles>"
cd "$(dirname "$0")/.."
for lib in $1; do
  for pr in 0 1; do
    tiles=$2; [ $pr = 1 ] && tiles=$3
    if [ "$lib" = "default" ]; then unset MF_LIB_OVERRIDE; else export MF_LIB_OVERRIDE=abl/$lib.so; fi
    timeout 300 python scripts/conv_sweep.py --precision $pr --tiles $tiles --reps 6 > gpurun_out/ab_${lib}_p$pr.txt 2>&1
    echo "$lib p$pr: $(tail -1 gpurun_out/ab_${lib}_p$pr.txt)"
  done
done
