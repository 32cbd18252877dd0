#!/bin/bash
# round-4 very last GPU session (~5 min of budget): the grouped launch (mf_conv2d_f16x2_group) -- bit-equality tests, the same-process A/B on
# cfg2, and the PMC traffic of the tree as it is now (the conv sources were split into body files: new stamp), in the configuration the A/B decides
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04group
mkdir -p $O
cd $R
timeout 110 python -m pytest tests/test_kernels_gpu.py -q -x -k "two_convolutions_in_one_launch or grouped_conv_res" > $O/tests.txt 2>&1; T=$?; tail -3 $O/tests.txt
timeout 90 python scripts/group_ab.py $O/ab.json 4 > $O/ab.txt 2>&1; tail -5 $O/ab.txt
ADOPT=$(python -c "
import json,sys
try:
    j=json.load(open('$O/ab.json')); print(1 if (j['bit_identical'] and j['gain_pct']>=0.7 and $T==0) else 0)
except Exception: print(0)")
echo "tests rc=$T adopt=$ADOPT"
if [ "$ADOPT" = "1" ]; then
  MEDFUSION_GROUPED_CONV_RES=1 timeout 100 python -m pytest tests/test_parity_gpu.py -q -x -k "published_unet_and_decode or published_short_trajectory" > $O/parity_on.txt 2>&1; P=$?; tail -2 $O/parity_on.txt
  if [ $P -ne 0 ]; then ADOPT=0; fi
fi
echo "adopt=$ADOPT" | tee $O/adopt.txt
if [ "$ADOPT" = "1" ]; then export MEDFUSION_GROUPED_CONV_RES=1; fi
timeout 120 bash scripts/pmc_bench_traffic.sh gpurun_out/r04group/pmc_traffic > $O/pmc_traffic.txt 2>&1; tail -4 $O/pmc_traffic.txt
rm -f $O/pmc_traffic/*_counter_collection.csv $O/pmc_traffic/*kernel_trace.csv $O/pmc_traffic/*agent_info*
