#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05s18
mkdir -p $O
cd $R
timeout 1200 python scripts/plan_tune.py profiles/r04_conv_sweep_b8.txt --batch 8 --top 3 --reps 2 > $O/plan_tune_b8.txt 2>&1; tail -12 $O/plan_tune_b8.txt | cut -c1-200
timeout 1200 python scripts/plan_tune.py profiles/r04_conv_sweep_planner_vs_best.txt --batch 16 --top 3 --reps 2 > $O/plan_tune_b16.txt 2>&1; tail -12 $O/plan_tune_b16.txt | cut -c1-200
