#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05s13
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_winograd_gpu.py -x -q -s > $O/tests_wino.txt 2>&1; tail -4 $O/tests_wino.txt
# same-process A/Bs on the cfg2 step: conv_res in the component GEMM's grid (MEDFUSION_WINOGRAD_GROUP), non-temporal V stores (MF_WINO_VSTORE)
for v in "0 0" "1 0" "0 1" "1 1"; do set -- $v
  MEDFUSION_WINOGRAD_GROUP=$1 MF_WINO_VSTORE=$2 timeout 600 python scripts/wino_ab.py $O/ab_g$1_nt$2.json 2 > $O/ab_g$1_nt$2.txt 2>&1; echo "group=$1 ntstore=$2: $(grep 'winograd  ' $O/ab_g$1_nt$2.txt)"
done
