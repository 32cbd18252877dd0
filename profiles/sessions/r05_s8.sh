#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05s8
mkdir -p $O
cd $R
timeout 600 python -m pytest tests/test_winograd_gpu.py -x -q 2>&1 | tail -3
timeout 1200 python scripts/wino_sweep.py --batch 16 --json $O/wino_b16.json > $O/wino_sweep_b16.txt 2>&1; tail -24 $O/wino_sweep_b16.txt
timeout 1200 python scripts/wino_sweep.py --batch 8 --json $O/wino_b8.json > $O/wino_sweep_b8.txt 2>&1; tail -3 $O/wino_sweep_b8.txt
timeout 1200 python scripts/wino_sweep.py --batch 8 --latent 64 --json $O/wino_l64.json > $O/wino_sweep_l64.txt 2>&1; tail -3 $O/wino_sweep_l64.txt
MEDFUSION_WINOGRAD_TABLE=$O/wino_b16.json timeout 600 python scripts/wino_ab.py $O/wino_ab_b16.json 3 > $O/wino_ab_b16.txt 2>&1; tail -6 $O/wino_ab_b16.txt
