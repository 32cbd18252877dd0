#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05s6
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_winograd_gpu.py -x -q -s > $O/tests_wino.txt 2>&1; tail -25 $O/tests_wino.txt
export MEDFUSION_WINOGRAD_TABLE=$R/scripts/wino_b16_fusedin.json
timeout 600 python scripts/wino_ab.py $O/wino_ab_b16_tail.json 3 > $O/wino_ab_b16_tail.txt 2>&1; tail -6 $O/wino_ab_b16_tail.txt
timeout 600 python scripts/wino_loop_prof.py > $O/loop_prof.txt 2>&1; tail -22 $O/loop_prof.txt
