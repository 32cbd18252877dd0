#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05s2
mkdir -p $O/prof
cd $R
export MEDFUSION_WINOGRAD_TABLE=$R/scripts/wino_b16.json
timeout 600 python scripts/wino_loop_prof.py > $O/loop_prof.txt 2>&1; cat $O/loop_prof.txt | tail -30
cd /tmp && export TMPDIR=/tmp
WINO_MODES=1 timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof -o wino --output-format csv -- python $R/scripts/wino_loop_prof.py trace > $O/prof.log 2>&1
cp $O/prof/wino_kernel_stats.csv $O/wino_kernel_stats.csv; rm -rf $O/prof
head -30 $O/wino_kernel_stats.csv | cut -c1-200
