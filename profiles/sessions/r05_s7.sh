#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05s7
mkdir -p $O
export MEDFUSION_WINOGRAD_TABLE=$R/scripts/wino_b16_fusedin.json
cd /tmp && export TMPDIR=/tmp
WINO_MODES=1 timeout 600 rocprofv3 --kernel-trace -d $O/prof1 -o t --output-format csv -- python $R/scripts/wino_loop_prof.py trace > $O/prof1.log 2>&1
python $R/scripts/trace_iteration.py $O/prof1/t_kernel_trace.csv $O/iteration_mode1.txt
rm -rf $O/prof1
cd $R; timeout 600 python -m pytest tests/test_winograd_gpu.py -x -q 2>&1 | tail -3
