#!/usr/bin/env python3
"""Same-process A/B of the fused conv + GroupNorm-apply launch (mf_conv2d_f16x2_gn_apply) on the benchmarked workload: cfg2 steps with the
form off / on from 16^2 pixels per sample / on from 32^2 only, interleaved so that clock and box drift hit every variant alike."""
import sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch
import medfusion_amd as M
from medfusion_amd import kernels as K, published as P

dev = torch.device("cuda:0")
pipe = P.build_published_pipeline(dev, None)
VARIANTS = [("two launches", True, 256), ("fused from 16x16", False, 256), ("fused from 32x32", False, 1024), ("fused everywhere", False, 1)]
ROUNDS, STEPS = 4, 3


def run(steps, seed):
    for k in range(steps):
        pipe.sample(16, (8, 32, 32), steps=150, use_ddim=True, noise=M.PhiloxDeviceNoise(seed + k))
    torch.cuda.synchronize()


res = {v[0]: [] for v in VARIANTS}
for r in range(ROUNDS):
    for name, off, minhw in VARIANTS:
        K.Rendezvous.disabled, K.FUSE_MIN_HW = off, minhw
        for m in pipe.modules():          # (the per-shape decision is cached on the Conv holders: drop it when the policy changes)
            if hasattr(m, "_descs"):
                m._descs = {k: v for k, v in m._descs.items() if not (isinstance(k, tuple) and k and k[0] == "fused")}
        run(1, 100)
        t0 = time.perf_counter()
        run(STEPS, 200 + 10 * r)
        res[name].append((time.perf_counter() - t0) / STEPS * 1e3)
print(f"cfg2 (B = 16, 150 DDIM iterations + decode), ms per step over {ROUNDS} interleaved rounds of {STEPS} steps:")
base = sum(res["two launches"]) / ROUNDS
for name, _, _ in VARIANTS:
    v = res[name]
    print(f"  {name:18s} " + " ".join(f"{x:7.2f}" for x in v) + f"   mean {sum(v) / len(v):7.2f} ms = {16e3 / (sum(v) / len(v)):6.2f} images/s  ({100 * (base / (sum(v) / len(v)) - 1):+.2f} % vs two launches)")
