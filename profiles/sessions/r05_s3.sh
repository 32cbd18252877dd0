#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05s3
mkdir -p $O
export MEDFUSION_WINOGRAD_TABLE=$R/scripts/wino_b16.json
cd /tmp && export TMPDIR=/tmp
for m in 0 1; do
  WINO_MODES=$m timeout 600 rocprofv3 --kernel-trace -d $O/prof$m -o t --output-format csv -- python $R/scripts/wino_loop_prof.py trace > $O/prof$m.log 2>&1
  python $R/scripts/trace_iteration.py $O/prof$m/t_kernel_trace.csv $O/iteration_mode$m.txt
  rm -rf $O/prof$m
done
tail -3 $O/iteration_mode0.txt $O/iteration_mode1.txt
