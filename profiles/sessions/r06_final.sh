#!/bin/bash
# round-6 final GPU session: the full GPU suite (with the [measured] lines), smoke, PMC traffic of the final tree (stamped), rocprofv3 kernel stats + idle-time
# analysis of the bench command, the driver-style bench line.  Everything lands in gpurun_out/r06final3/; the summaries are copied to profiles/ by hand.
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06final3
mkdir -p $O/prof
cd $R
timeout 2700 python -m pytest tests -m gpu -q -s > $O/tests_all.txt 2>&1; tail -4 $O/tests_all.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 1500 bash scripts/pmc_bench_traffic.sh gpurun_out/r06final3/pmc_traffic > $O/pmc_traffic.txt 2>&1; tail -9 $O/pmc_traffic.txt
rm -f $O/pmc_traffic/*_counter_collection.csv $O/pmc_traffic/*kernel_trace.csv $O/pmc_traffic/*agent_info*
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats -d $O/prof -o bench --output-format csv -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-alt-path --no-roofline --no-other-workloads > $O/prof_bench.json 2> $O/prof_bench.err
python $R/scripts/trace_gaps.py $O/prof/bench_kernel_trace.csv $O/trace_gaps.txt | head -6
cp $O/prof/bench_kernel_stats.csv $O/kernel_stats.csv; rm -rf $O/prof
cd $R
timeout 1500 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err; head -c 300 $O/bench.json; echo
