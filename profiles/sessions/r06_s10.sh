#!/bin/bash
# round-6 session 10: do the 8 x 8-level component GEMMs care whether their weights come from HBM or from the Infinity Cache?  The same GEMM + tail sequence
# cycling through 1.2 GB of weight copies (cold: what the loop sees) and through 2 copies (134 MB: resident in the 256 MB memory-side cache)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06s10
mkdir -p $O
cd $R
for mb in 1200 1; do
  timeout 600 python scripts/wino_sweep.py --batch 16 --reps 8 --cold-mb $mb --only "R1024.c0" > $O/wino_cold$mb.txt 2>&1; grep "R1024.c0" $O/wino_cold$mb.txt
  timeout 600 python scripts/wino_sweep.py --batch 16 --reps 8 --cold-mb $mb --only "in16 R512.c0" >> $O/wino_cold$mb.txt 2>&1; grep "R512.c0" $O/wino_cold$mb.txt
done
