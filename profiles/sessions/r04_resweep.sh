#!/bin/bash
# round-4 re-sweep of the planner table on the current kernels (the B = 16 pass of this session found the 8-wave 128x128 tile ahead on the
# split-K shapes: +3.1 % on the cfg2 step): every workload's shapes, entries emitted where the sweep beats the planner by > 1.5 %
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04sw
mkdir -p $O
cd $R
rm -f $O/table_*.inc
timeout 900 python scripts/conv_sweep.py --precision 5 --reps 8 --batch 16 --vae-batch 16 --emit-table $O/table_b16.inc > $O/sweep_b16.txt 2>&1; tail -1 $O/sweep_b16.txt
timeout 900 python scripts/conv_sweep.py --precision 5 --reps 8 --batch 32 --vae-batch 16 --only R --emit-table $O/table_b32.inc > $O/sweep_b32.txt 2>&1; tail -1 $O/sweep_b32.txt
timeout 900 python scripts/conv_sweep.py --precision 5 --reps 8 --batch 32 --vae-batch 16 --only "n" --emit-table $O/table_b32b.inc > $O/sweep_b32b.txt 2>&1; tail -1 $O/sweep_b32b.txt
timeout 900 python scripts/conv_sweep.py --precision 5 --reps 8 --batch 8 --vae-batch 8 --emit-table $O/table_b8.inc > $O/sweep_b8.txt 2>&1; tail -1 $O/sweep_b8.txt
timeout 900 python scripts/conv_sweep.py --precision 5 --reps 8 --batch 8 --vae-batch 8 --latent 64 --emit-table $O/table_l64.inc > $O/sweep_l64.txt 2>&1; tail -1 $O/sweep_l64.txt
for f in $O/table_*.inc; do echo "== $f"; sort -u $f; done
