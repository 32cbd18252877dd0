#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05s5
mkdir -p $O
cd $R
timeout 1200 python scripts/wino_sweep.py --batch 16 --reps 8 --json $O/wino_b16.json > $O/wino_sweep_b16.txt 2>&1; grep -v "component GEMM" $O/wino_sweep_b16.txt | tail -24
MEDFUSION_WINOGRAD_TABLE=$O/wino_b16.json timeout 600 python scripts/wino_ab.py $O/wino_ab_b16.json 3 > $O/wino_ab_b16.txt 2>&1; tail -6 $O/wino_ab_b16.txt
python scripts/wino_sweep.py --batch 16 --reps 8 --fused-input --quick --json $O/wino_b16_fusedin.json > /dev/null 2>&1
MEDFUSION_WINOGRAD_TABLE=$O/wino_b16_fusedin.json timeout 600 python scripts/wino_ab.py $O/wino_ab_b16_all.json 2 > $O/wino_ab_b16_all.txt 2>&1; tail -6 $O/wino_ab_b16_all.txt
cat $O/wino_b16.json; echo; cat $O/wino_b16_fusedin.json
