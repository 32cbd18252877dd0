#!/bin/bash
# round-6 session 1: where the tree stands on this round's box + the cheap experiments that decide the round's plan
#  (a) cfg2 bench line (short), (b) the reference's bulk workload as shipped (chunks of 200, tail 69), (c) the Winograd / direct sweep at
#  batches the tables do not hold (12, 24, 69, 200), (d) the fused-output-Winograd proxy for the 32x32 level, (e) matrix-pipe bursts from part of the chip
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06s1
mkdir -p $O
cd $R
timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-alt-path --no-other-workloads > $O/bench.json 2> $O/bench.err; head -c 400 $O/bench.json; echo
hipcc --offload-arch=gfx950 -O3 scripts/mfma_power_probe.hip -o /tmp/mfma_power_probe 2>/dev/null && timeout 120 /tmp/mfma_power_probe part > $O/power_part.txt 2>&1; tail -5 $O/power_part.txt
timeout 600 python scripts/wino32_proxy.py > $O/wino32_proxy.txt 2>&1; cat $O/wino32_proxy.txt | tail -12
timeout 900 python scripts/sample_dataset.py --synthetic --sample-batch 200 --n-samples 469 --steps-list 150 --labels No_Cardiomegaly:0 --no-files --compare-no-egress > $O/bulk200.txt 2>&1; tail -2 $O/bulk200.txt
for b in 12 24 69 200; do
  timeout 900 python scripts/wino_sweep.py --batch $b --reps 4 --cold-mb 600 > $O/wino_sweep_b$b.txt 2>&1; tail -3 $O/wino_sweep_b$b.txt
done
