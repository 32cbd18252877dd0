#!/bin/bash
# round-4 GPU session 2: the fused conv + GroupNorm-apply launch -- unit tests, A/B bench, goldens
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04s2
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_kernels_gpu.py -m gpu -q -x -k "gn_apply_in_one_launch or fused_refuses" > $O/tests_fused.txt 2>&1; tail -15 $O/tests_fused.txt
timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-alt-path --no-other-workloads > $O/bench_fused.json 2> $O/bench_fused.err; python - <<'PY'
import json
for tag in ("fused",):
    try:
        d = json.load(open(f"gpurun_out/r04s2/bench_{tag}.json"))
        print(tag, d["value"], d["ms_per_step"], d["roofline"]["families_ms"])
    except Exception as e:
        print(tag, "FAILED", e)
PY
tail -3 $O/bench_fused.err
MEDFUSION_FUSED_APPLY=0 timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-alt-path --no-other-workloads > $O/bench_unfused.json 2> $O/bench_unfused.err; python - <<'PY'
import json
try:
    d = json.load(open("gpurun_out/r04s2/bench_unfused.json"))
    print("unfused", d["value"], d["ms_per_step"], d["roofline"]["families_ms"])
except Exception as e:
    print("unfused FAILED", e)
PY
timeout 900 python -m pytest tests/test_parity_gpu.py -m gpu -q -x -k "blocks_golden or unet_tiny or sample_tiny or cfg1 or vae_tiny or hipgraph or full_size_properties" > $O/tests_golden.txt 2>&1; tail -6 $O/tests_golden.txt
timeout 600 python -m pytest tests/test_kernels_gpu.py -m gpu -q -x -k "timeout_falls_back" > $O/tests_fault.txt 2>&1; tail -8 $O/tests_fault.txt
