#!/bin/bash
mkdir -p gpurun_out
cd /root/repo
cp medfusion_amd/libmedfusion_hip.so /tmp/keep.so
for v in 0 1 2 4 3 7; do
  cp abl/lib$v.so medfusion_amd/libmedfusion_hip.so
  for z in "" "--zeros"; do
    for t in "31 8 16,8,8,1024,1024,1024,3,1,0" "33 1 16,32,32,256,0,256,3,1,0" "52 1 16,32,32,256,0,256,3,1,0"; do
      set -- $t
      echo "ablate $v $z: $(timeout 120 python scripts/conv_one.py --precision 5 --tile $1 --splitk $2 --shape $3 --reps 30 $z 2>&1 | tail -1)"
    done
  done
done | tee gpurun_out/r02v_ablate.txt
cp /tmp/keep.so medfusion_amd/libmedfusion_hip.so
