#!/bin/bash
# round-4 GPU session 3: diagnose the fused conv + GroupNorm-apply launch (all unit cases, the phase timeline of the tail)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04s3
mkdir -p $O
cd $R
timeout 1200 python -m pytest tests/test_kernels_gpu.py -m gpu -q -k "gn_apply_in_one_launch or fused_refuses or timeout_falls_back" > $O/tests_fused.txt 2>&1; grep -a "differ:\|passed\|failed\|FAILED\|Error" $O/tests_fused.txt | cut -c1-900
MEDFUSION_LIB=$R/medfusion_amd/csrc/build/variants/libmedfusion_hip_stamp.so timeout 300 python scripts/conv_timeline.py > $O/timeline_unfused.txt 2>&1; cat $O/timeline_unfused.txt
MEDFUSION_LIB=$R/medfusion_amd/csrc/build/variants/libmedfusion_hip_stamp.so timeout 300 python scripts/conv_timeline.py --fused > $O/timeline_fused.txt 2>&1; cat $O/timeline_fused.txt
