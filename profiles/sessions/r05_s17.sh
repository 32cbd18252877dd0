#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05s17
mkdir -p $O
cd $R
timeout 2400 python -m pytest tests -m gpu -q > $O/tests_all.txt 2>&1; tail -4 $O/tests_all.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
