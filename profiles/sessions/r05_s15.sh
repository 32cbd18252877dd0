#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05s15
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
WINO_B=8 WINO_MODES=1 timeout 600 rocprofv3 --kernel-trace -d $O/prof -o t --output-format csv -- python $R/scripts/wino_loop_prof.py trace > $O/prof.log 2>&1
python $R/scripts/trace_iteration.py $O/prof/t_kernel_trace.csv $O/iteration_b8.txt
rm -rf $O/prof
tail -2 $O/iteration_b8.txt
