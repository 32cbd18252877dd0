#!/bin/bash
# round-6 session 4: same-box A/Bs on cfg2 -- (a) the round-5 planner / table against the round-6 model / rule (must be equal at B = 16: every shape
# is in the table), (b) the XCD-aware (sample, group) order of the Winograd tail; (c) the component GEMM's tiles at B = 200 (is the old model's pick good there?)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06s4
mkdir -p $O
cd $R
timeout 1500 python scripts/env_ab.py --rounds 3 "MF_PLAN_MODEL=0 MF_WINO_RULE=0 MF_WINO_TAIL_MAP=0" "MF_WINO_TAIL_MAP=0" "" > $O/ab_cfg2.txt 2>&1; cat $O/ab_cfg2.txt
timeout 900 python scripts/wino_sweep.py --batch 200 --reps 4 --cold-mb 600 --tiles --only "R512.c0" > $O/wino_tiles_b200.txt 2>&1; tail -4 $O/wino_tiles_b200.txt
timeout 900 python scripts/wino_sweep.py --batch 200 --reps 4 --cold-mb 600 --tiles --only "R1024.c0" >> $O/wino_tiles_b200.txt 2>&1; tail -4 $O/wino_tiles_b200.txt
timeout 600 python -m pytest tests/test_winograd_gpu.py -x -q -k "tail" > $O/tests_tail.txt 2>&1; tail -3 $O/tests_tail.txt
