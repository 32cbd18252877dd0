#!/bin/bash
# round-3 artefact session (GPU box): rocprofv3 kernel stats + gap analysis of the bench command, PMC passes of the conv kernel (the two
# most frequent tiles), HBM traffic of the bench command (PMC), the bench line itself, the other workloads, the micro benches, the
# packed-fp32 erratum repro, the full GPU suite, smoke.  Everything lands in gpurun_out/r03z/; the summaries are copied to profiles/ by hand.
set -x
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03z
mkdir -p $O/prof
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats -d $O/prof -o bench --output-format csv -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-alt-path --no-roofline --no-other-workloads > $O/prof_bench.json 2> $O/prof_bench.err
python $R/scripts/trace_gaps.py $O/prof/bench_kernel_trace.csv $O/trace_gaps.txt | head -12
cp $O/prof/bench_kernel_stats.csv $O/kernel_stats.csv
rm -f $O/prof/bench_kernel_trace.csv $O/prof/*agent_info* $O/prof/*domain_stats*
cd $R
timeout 900 bash scripts/pmc_conv.sh gpurun_out/r03z/pmc_t52 52 1 16,32,32,256,0,256,3,1,0 all 5 > $O/pmc_t52.txt 2>&1
timeout 900 bash scripts/pmc_conv.sh gpurun_out/r03z/pmc_t31 31 8 16,8,8,1024,1024,1024,3,1,0 all 5 > $O/pmc_t31.txt 2>&1
rm -f $O/pmc_t*/*_counter_collection.csv $O/pmc_t*/*kernel_trace.csv $O/pmc_t*/*agent_info*
timeout 2400 bash scripts/pmc_bench_traffic.sh gpurun_out/r03z/pmc_traffic > $O/pmc_traffic.txt 2>&1; tail -8 $O/pmc_traffic.txt
rm -f $O/pmc_traffic/*_counter_collection.csv $O/pmc_traffic/*kernel_trace.csv $O/pmc_traffic/*agent_info*
timeout 1200 python bench.py --steps 3 --warmup 1 > $O/bench.json 2> $O/bench.err; head -c 300 $O/bench.json
timeout 2400 bash scripts/other_workloads.sh gpurun_out/r03z/other_workloads.txt > /dev/null 2>&1; cat $O/other_workloads.txt
timeout 300 python scripts/gn_apply_bench.py > $O/gn_apply_bench.txt 2>&1
timeout 300 python scripts/conv_fixed_cost.py --tiles 52,53,33 > $O/conv_fixed_cost.txt 2>&1
timeout 300 python scripts/enqueue_time.py > $O/enqueue_time.txt 2>&1
(hipcc --offload-arch=gfx950 -O2 -o /tmp/pk_repro_min scripts/pk_repro_min.hip && timeout 120 /tmp/pk_repro_min) > $O/pk_repro_min.txt 2>&1; tail -5 $O/pk_repro_min.txt
timeout 2400 python -m pytest tests -m gpu -q > $O/tests_all.txt 2>&1; tail -4 $O/tests_all.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1; tail -2 $O/smoke.txt
