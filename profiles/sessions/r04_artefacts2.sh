#!/bin/bash
# round-4 artefact refresh after the plan-table re-sweep: rocprofv3 kernel stats + gap analysis of the bench command, HBM traffic (PMC, stamped),
# PMC passes of the two dominant tiles (52 at 32^2 without split-K, 34 at 16^2 with a 2-way in-launch split-K)
set -x
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04y
mkdir -p $O/prof
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats -d $O/prof -o bench --output-format csv -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-alt-path --no-roofline --no-other-workloads > $O/prof_bench.json 2> $O/prof_bench.err
python $R/scripts/trace_gaps.py $O/prof/bench_kernel_trace.csv $O/trace_gaps.txt | head -8
cp $O/prof/bench_kernel_stats.csv $O/kernel_stats.csv
rm -f $O/prof/bench_kernel_trace.csv $O/prof/*agent_info* $O/prof/*domain_stats*
cd $R
timeout 2400 bash scripts/pmc_bench_traffic.sh gpurun_out/r04y/pmc_traffic > $O/pmc_traffic.txt 2>&1; tail -8 $O/pmc_traffic.txt
rm -f $O/pmc_traffic/*_counter_collection.csv $O/pmc_traffic/*kernel_trace.csv $O/pmc_traffic/*agent_info*
timeout 900 bash scripts/pmc_conv.sh gpurun_out/r04y/pmc_t34 34 2 16,16,16,512,0,512,3,1,0 core 5 > $O/pmc_t34.txt 2>&1
rm -f $O/pmc_t*/*_counter_collection.csv $O/pmc_t*/*kernel_trace.csv $O/pmc_t*/*agent_info*
cat $O/pmc_t34/summary.csv
