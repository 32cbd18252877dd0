#!/bin/bash
# round-6 session 12: the small-Cin edge convolution (UNet in_conv, VAE inc / inc_dec) on the round-6 kernel: bit-equality with the round-5 one, the conv
# tests, and the same-box A/B on cfg2 (MF_SMALLCIN=0 / 1)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06s12
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_kernels_gpu.py -q -x -k "conv" > $O/tests_conv.txt 2>&1; tail -3 $O/tests_conv.txt
timeout 1200 python scripts/env_ab.py --rounds 3 "MF_SMALLCIN=0" "MF_SMALLCIN=1" > $O/ab_smallcin.txt 2>&1; cat $O/ab_smallcin.txt
timeout 300 python - > $O/prof_smallcin.txt 2>&1 <<'PY'
import torch, medfusion_amd as M
from medfusion_amd import kernels as K, published as P
pipe = P.build_published_pipeline(torch.device("cuda:0"), None)
pipe.sample(16, (8, 32, 32), steps=10, use_ddim=True, noise=M.PhiloxDeviceNoise(1))
with K.prof() as p:
    pipe.sample(16, (8, 32, 32), steps=10, use_ddim=True, noise=M.PhiloxDeviceNoise(2))
for r in K.prof_rows("conv_direct"): print(r["kernel"][:70], r["launches"], round(r["ms"] / r["launches"] * 1e3, 2), "us")
PY
cat $O/prof_smallcin.txt | tail -5
