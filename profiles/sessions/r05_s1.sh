#!/bin/bash
# round-5 GPU session 1: the Winograd prototype -- correctness, per-shape sweep against the direct form, same-process cfg2 A/B
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05s1
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_winograd_gpu.py -x -q -s > $O/tests_wino.txt 2>&1; tail -5 $O/tests_wino.txt
timeout 900 python scripts/wino_sweep.py --batch 16 --json $O/wino_b16.json > $O/wino_sweep_b16.txt 2>&1; tail -22 $O/wino_sweep_b16.txt
MEDFUSION_WINOGRAD_TABLE=$O/wino_b16.json timeout 600 python scripts/wino_ab.py $O/wino_ab_b16.json 3 > $O/wino_ab_b16.txt 2>&1; tail -8 $O/wino_ab_b16.txt
