#!/bin/bash
# round-6 session 5: the component GEMM of the Winograd form over (tile, split-K) at every batch on file (8 / 12 / 16 / 24 / 32 / 69 / 200): the data the
# fitted cost model is checked against before it replaces the round-5 model for the GEMMs too; + the fused-output proxy at larger batches
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06s5
mkdir -p $O
cd $R
for b in 16 8 32 12 24 69 200; do
  timeout 1500 python scripts/wino_sweep.py --batch $b --reps 4 --cold-mb 600 --tiles > $O/wino_tiles_b$b.txt 2>&1; tail -1 $O/wino_tiles_b$b.txt
done
for b in 32 64 200; do timeout 600 python scripts/wino32_proxy.py --batch $b > $O/wino32_proxy_b$b.txt 2>&1; tail -3 $O/wino32_proxy_b$b.txt; done
