#!/bin/bash
# round-4 GPU session 4: the fused launch after the VALU diet (fast Swish, no clamp) and the bit_cast fix
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04s4
mkdir -p $O
cd $R
timeout 1200 python -m pytest tests/test_kernels_gpu.py -m gpu -q -k "gn_apply_in_one_launch or fused_refuses or timeout_falls_back or groupnorm or gn_apply" > $O/tests_fused.txt 2>&1; grep -a "differ:\|passed\|failed\|FAILED\|Error" $O/tests_fused.txt | cut -c1-700
MEDFUSION_LIB=$R/medfusion_amd/csrc/build/variants/libmedfusion_hip_stamp.so timeout 300 python scripts/conv_timeline.py --fused > $O/timeline_fused.txt 2>&1; cat $O/timeline_fused.txt
for hw in 256 64 100000; do
  MEDFUSION_FUSE_MIN_HW=$hw timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-alt-path --no-other-workloads > $O/bench_minhw$hw.json 2> $O/bench_minhw$hw.err
  python - $hw <<'PY'
import json, sys
try:
    d = json.load(open(f"gpurun_out/r04s4/bench_minhw{sys.argv[1]}.json"))
    print("min_hw", sys.argv[1], d["value"], d["ms_per_step"], d["roofline"]["families_ms"])
except Exception as e:
    print("min_hw", sys.argv[1], "FAILED", e)
PY
done
timeout 900 python -m pytest tests/test_parity_gpu.py -m gpu -q -x -k "blocks_golden or unet_tiny or sample_tiny or cfg1 or vae_tiny or hipgraph or full_size_properties or published_unet" > $O/tests_golden.txt 2>&1; tail -6 $O/tests_golden.txt
timeout 300 python scripts/gn_apply_bench.py > $O/gn_apply_bench.txt 2>&1; tail -12 $O/gn_apply_bench.txt
