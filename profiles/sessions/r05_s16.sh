#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05s16
mkdir -p $O
cd $R
MEDFUSION_WINOGRAD=2 timeout 2400 python -m pytest tests -m gpu -q > $O/tests_wino_everywhere.txt 2>&1; tail -15 $O/tests_wino_everywhere.txt | cut -c1-300
