#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05s11
mkdir -p $O
cd $R
timeout 1800 python -m pytest tests/test_parity_gpu.py -x -q -s > $O/tests_parity.txt 2>&1; grep "measured" $O/tests_parity.txt | grep -i "trained\|slack" | cut -c1-220; tail -3 $O/tests_parity.txt
