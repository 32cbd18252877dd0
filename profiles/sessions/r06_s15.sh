#!/bin/bash
# round-6 session 15: the opt-in Winograd form on the fp32 MFMA arithmetic (test + step time), the bench line with the new other_conv_arithmetic entries
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06s15
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_parity_gpu.py -q -x -s -k "fp32_mfma_with_the_winograd" > $O/tests_mfma.txt 2>&1; tail -4 $O/tests_mfma.txt
timeout 900 python bench.py --steps 3 --warmup 1 --no-other-workloads --no-cpu-baseline > $O/bench.json 2> $O/bench.err; python - <<'PY'
import json
j = json.loads(open("gpurun_out/r06s15/bench.json").read())
print(j["value"])
for a in j["other_conv_arithmetic"]:
    print({k: v for k, v in a.items() if k != "arithmetic"})
PY
