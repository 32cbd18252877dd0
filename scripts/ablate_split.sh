#!/bin/bash
# Timing ablation of the split-mode conv kernel.  Step 1 (build container): scripts/ablate_split.sh build   -> abl/lib_<bits>.so
# Step 2 (GPU box): scripts/ablate_split.sh run
cd "$(dirname "$0")/.."
BITS=${BITS:-"0 1 2 4 8 16 3 7 15 31 32 63 64 68"}
if [ "$1" = "build" ]; then
  mkdir -p abl
  for a in $BITS; do
    ( /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -fno-gpu-rdc -ffp-contract=off -mllvm -pragma-unroll-threshold=1000000 \
      -DMF_ABLATE=$a $EXTRA medfusion_amd/csrc/*.hip -o abl/lib_$a.so 2>/dev/null && echo built $a ) &
    if (( $(jobs -r | wc -l) >= 6 )); then wait -n; fi
  done
  wait
else
  for t in "8 1" "9 2"; do set -- $t
    for a in $BITS; do
      echo -n "tile $1 ablate $a: "; MF_LIB_OVERRIDE=abl/lib_$a.so python scripts/conv_one.py --shape 16,32,32,512,0,256,3,1,0 --tile $1 --splitk $2 --precision 1 --reps 20 2>&1 | tail -1
    done
  done
fi
