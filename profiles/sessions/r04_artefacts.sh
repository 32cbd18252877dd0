#!/bin/bash
# round-4 artefact session (GPU box): rocprofv3 kernel stats + gap analysis of the bench command, PMC passes of the dominant conv tile, HBM
# traffic of the bench command (PMC, stamped with the conv sources' hash), the bench line itself (every BASELINE config + CPU baseline), the
# full GPU suite, smoke.  Everything lands in gpurun_out/r04z/; the summaries are copied to profiles/ by hand.
set -x
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04z
mkdir -p $O/prof
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats -d $O/prof -o bench --output-format csv -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-alt-path --no-roofline --no-other-workloads > $O/prof_bench.json 2> $O/prof_bench.err
python $R/scripts/trace_gaps.py $O/prof/bench_kernel_trace.csv $O/trace_gaps.txt | head -12
cp $O/prof/bench_kernel_stats.csv $O/kernel_stats.csv
rm -f $O/prof/bench_kernel_trace.csv $O/prof/*agent_info* $O/prof/*domain_stats*
cd $R
timeout 900 bash scripts/pmc_conv.sh gpurun_out/r04z/pmc_t52 52 1 16,32,32,256,0,256,3,1,0 all 5 > $O/pmc_t52.txt 2>&1
rm -f $O/pmc_t*/*_counter_collection.csv $O/pmc_t*/*kernel_trace.csv $O/pmc_t*/*agent_info*
timeout 2400 bash scripts/pmc_bench_traffic.sh gpurun_out/r04z/pmc_traffic > $O/pmc_traffic.txt 2>&1; tail -8 $O/pmc_traffic.txt
rm -f $O/pmc_traffic/*_counter_collection.csv $O/pmc_traffic/*kernel_trace.csv $O/pmc_traffic/*agent_info*
timeout 1500 python bench.py --steps 5 --warmup 1 > $O/bench.json 2> $O/bench.err; head -c 300 $O/bench.json
timeout 2400 python -m pytest tests -m gpu -q > $O/tests_all.txt 2>&1; tail -4 $O/tests_all.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1; tail -2 $O/smoke.txt
