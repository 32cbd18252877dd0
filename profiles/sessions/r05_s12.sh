#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05s12
mkdir -p $O
cd $R
timeout 1500 python scripts/wino_sweep.py --batch 32 --json $O/wino_b32.json > $O/wino_sweep_b32.txt 2>&1; tail -3 $O/wino_sweep_b32.txt
timeout 900 python scripts/wino_sweep.py --batch 4 --json $O/wino_b4.json > $O/wino_sweep_b4.txt 2>&1; tail -3 $O/wino_sweep_b4.txt
timeout 900 python scripts/wino_sweep.py --batch 16 --latent 64 --json $O/wino_b16_l64.json > $O/wino_sweep_b16_l64.txt 2>&1; tail -3 $O/wino_sweep_b16_l64.txt
