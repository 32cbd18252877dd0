#!/usr/bin/env python3
"""Where the denoise loop's time goes with the Winograd form off / on: per-family device time (K.prof: every launch bracketed by events, eager
loop) of one 30-iteration loop at B = 16.  argv[1] = "trace": one loop per mode only (to run under rocprofv3 --kernel-trace --stats)."""
import os, sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch
import medfusion_amd as M
from medfusion_amd import blocks as BLK, kernels as K, published as P

dev = torch.device("cuda:0")
pipe = P.build_published_pipeline(dev, None)
B, ITS = int(os.environ.get("WINO_B", "16")), 30
modes = [int(m) for m in os.environ.get("WINO_MODES", "0,1").split(",")]
for mode in modes:
    BLK.WINOGRAD = mode
    pipe.sample(B, (8, 32, 32), steps=6, use_ddim=True, noise=M.PhiloxDeviceNoise(1), decode=False)
    torch.cuda.synchronize()
    if len(sys.argv) > 1 and sys.argv[1] == "trace":
        pipe.sample(B, (8, 32, 32), steps=ITS, use_ddim=True, noise=M.PhiloxDeviceNoise(2), decode=False)
        torch.cuda.synchronize()
        continue
    with K.prof() as p:
        pipe.sample(B, (8, 32, 32), steps=ITS, use_ddim=True, noise=M.PhiloxDeviceNoise(2), decode=False)
    t = p.table()
    tot = sum(v[0] for v in t.values())
    print(f"WINOGRAD = {mode}: {tot / ITS * 1e3:.0f} us of kernel time per iteration")
    for name, (ms, n, fl, by, ex) in sorted(t.items(), key=lambda kv: -kv[1][0]):
        print(f"  {name:14s} {ms / ITS * 1e3:8.1f} us/iteration  {n / ITS:6.1f} launches/iteration  avg {ms / n * 1e3:6.1f} us")
