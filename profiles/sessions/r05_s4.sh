#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05s4
mkdir -p $O
cd $R
timeout 900 python scripts/wino_sweep.py --batch 16 --reps 6 --only "R1024" --cold-mb 1300 > $O/wino_sweep_cold_8.txt 2>&1; tail -8 $O/wino_sweep_cold_8.txt
timeout 900 python scripts/wino_sweep.py --batch 16 --reps 6 --only "in8/mid R1024.c0" --cold-mb 1300 --with-input > $O/wino_sweep_cold_8_in.txt 2>&1; tail -4 $O/wino_sweep_cold_8_in.txt
timeout 900 python scripts/wino_sweep.py --batch 16 --reps 6 --only "in16 R512.c0" --cold-mb 1300 > $O/wino_sweep_cold_16.txt 2>&1; tail -4 $O/wino_sweep_cold_16.txt
