#!/bin/bash
# round-6 session 2: the direct fp16-pair convolution over (tile, split-K) at batches the plan table does not hold (12, 24, 69, 200; VAE at the same batch):
# what a batch-general plan rule has to reproduce
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06s2
mkdir -p $O
cd $R
for b in 200 69 24 12; do
  timeout 1500 python scripts/conv_sweep.py --precision 5 --batch $b --vae-batch $b --reps 4 > $O/conv_sweep_b$b.txt 2>&1; tail -1 $O/conv_sweep_b$b.txt
done
