#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05s14
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_winograd_gpu.py tests/test_kernels_gpu.py -x -q -k "wino or conv_f16x2" > $O/tests.txt 2>&1; tail -3 $O/tests.txt
timeout 900 python scripts/wino_sweep.py --batch 16 --only "in8/mid R1024.c0" --tiles > $O/sweep_tiles_8.txt 2>&1; tail -4 $O/sweep_tiles_8.txt | cut -c1-220
timeout 900 python scripts/wino_sweep.py --batch 16 --only "out8 R2048-1024.c0" --tiles > $O/sweep_tiles_8b.txt 2>&1; tail -3 $O/sweep_tiles_8b.txt | cut -c1-220
timeout 900 python scripts/wino_sweep.py --batch 16 --only "in16 R512.c0" --tiles > $O/sweep_tiles_16.txt 2>&1; tail -3 $O/sweep_tiles_16.txt | cut -c1-220
for v in 0 1; do
  MF_WINO_MSTORE=$v timeout 600 python scripts/wino_ab.py $O/ab_mnt$v.json 2 > $O/ab_mnt$v.txt 2>&1; echo "GEMM output nt=$v: $(grep 'winograd  ' $O/ab_mnt$v.txt)"
done
