#!/bin/bash
# round-6 session 8: the tile walk (MF_CONV_WALK: 0 pixel tiles fastest = round 5, 1 by operand size, 2 output-channel tiles fastest everywhere), same-box A/B on cfg2
# and at the bulk batch; the convolution tests on the default walk
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06s8
mkdir -p $O
cd $R
timeout 1500 python scripts/env_ab.py --rounds 3 "MF_CONV_WALK=0" "MF_CONV_WALK=1" "MF_CONV_WALK=2" > $O/ab_walk_cfg2.txt 2>&1; cat $O/ab_walk_cfg2.txt
timeout 1500 python scripts/env_ab.py --rounds 2 --batch 200 "MF_CONV_WALK=0" "MF_CONV_WALK=1" > $O/ab_walk_b200.txt 2>&1; cat $O/ab_walk_b200.txt
timeout 1500 python -m pytest tests/test_kernels_gpu.py tests/test_winograd_gpu.py -q -x > $O/tests_kernels.txt 2>&1; tail -3 $O/tests_kernels.txt
