#!/bin/bash
# the grouped launch as the default: pipeline-level bit equality at B = 16 / B = 8 (command list and eager), the graph / command-list test
# at B = 8, and the driver-style bench line
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04group2
mkdir -p $O
cd $R
timeout 80 python scripts/group_ab.py $O/check.json 0 > $O/check.txt 2>&1; echo "check rc=$?"; cat $O/check.txt | tail -4
timeout 70 python -m pytest tests/test_harness_gpu.py -q -x -k "cfg4_graph_replay" > $O/tests.txt 2>&1; echo "tests rc=$?"; tail -2 $O/tests.txt
timeout 150 python bench.py --gpus 1 --steps 10 --warmup 3 > $O/bench.json 2> $O/bench.err; head -c 230 $O/bench.json; echo
