#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05s10
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_winograd_gpu.py -x -q -s > $O/tests_wino.txt 2>&1; tail -4 $O/tests_wino.txt
timeout 900 python -m pytest tests/test_parity_gpu.py -x -q -s -k "trained_like or progress" > $O/tests_trained_like.txt 2>&1; grep "measured\|passed\|failed" $O/tests_trained_like.txt | cut -c1-260
timeout 900 python scripts/wino_sweep.py --batch 16 --only "R512" > $O/wino_sweep_prefetch.txt 2>&1; grep -v "^#\|^shape" $O/wino_sweep_prefetch.txt | cut -c1-140
timeout 900 python scripts/wino_sweep.py --batch 16 --only "R1024.c" > $O/wino_sweep_prefetch8.txt 2>&1; grep -v "^#\|^shape" $O/wino_sweep_prefetch8.txt | cut -c1-140
timeout 600 python scripts/wino_ab.py $O/wino_ab.json 3 > $O/wino_ab.txt 2>&1; tail -5 $O/wino_ab.txt
