#!/bin/bash
mkdir -p gpurun_out
cd /root/repo
timeout 900 python -m pytest tests/test_kernels_gpu.py -q -x -k "fused_groupnorm or split_mirror or residual_bound_slots" > gpurun_out/r02s_t_kernels.log 2>&1; tail -3 gpurun_out/r02s_t_kernels.log
for lib in new old new old; do
cp medfusion_amd/libmedfusion_hip.so.$lib medfusion_amd/libmedfusion_hip.so
timeout 900 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-alt-path > gpurun_out/r02s_bench_$lib.json 2> gpurun_out/r02s_bench_$lib.err; python - <<PY
import json
d=json.load(open('gpurun_out/r02s_bench_$lib.json'))
print("$lib", d['value'], d['roofline']['families_ms'])
PY
done
cp medfusion_amd/libmedfusion_hip.so.new medfusion_amd/libmedfusion_hip.so
