#!/usr/bin/env python3
"""The component GEMMs of the Winograd form on the exact bf16-triplet arithmetic over the implicit-GEMM kernel's (tile, split-K): GEMM (+ split-K reducer) time per
published shape at batch B, next to the planner's own pick (tile 0 / split-K 0) and to the direct 3x3 of the same arithmetic (device us, command-list replay)."""
import argparse, sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1])); sys.path.insert(0, str(Path(__file__).resolve().parent))
import torch
from medfusion_amd import kernels as K
from _devtime import device_us
from conv_sweep import unet_shapes

ap = argparse.ArgumentParser(); ap.add_argument("--batch", type=int, default=16); ap.add_argument("--reps", type=int, default=10); a = ap.parse_args()
dev = torch.device("cuda:0"); g = torch.Generator().manual_seed(0)
seen = set()
for name, n, h, w, c1, c2, co, k, st, ups, cnt in unet_shapes(a.batch):
    if k != 3 or st != 1 or ups or h > 16 or (h, c1 + c2, co) in seen:
        continue
    seen.add((h, c1 + c2, co))
    t = (h // 2) * (w // 2); cin = c1 + c2
    v = torch.randn((16 * n, 1, t, cin), generator=g).to(dev)
    u = K.split_conv_weight((torch.randn((16 * co, 1, 1, cin), generator=g) * 0.02).to(dev))
    x = torch.randn((n, h, w, cin), generator=g).to(dev)
    wd = K.split_conv_weight((torch.randn((co, 3, 3, cin), generator=g) * 0.02).to(dev))
    dd = K.make_conv_desc(n, h, w, cin, 0, co, 3, 1, 1, 0, precision=3)
    t_direct = device_us(lambda: K.conv2d(x, wd, None, dd), a.reps)[0]
    res = []
    for tile in (0, 1, 3, 7, 8, 9, 10):
        for sk in ((0,) if tile == 0 else (1, 2, 4)):
            d = K.make_conv_desc(16 * n, 1, t, cin, 0, co, 1, 1, 0, 3, tile_hint=tile, splitk_hint=sk, precision=3)
            try:
                us = device_us(lambda: K.conv2d(v, u, None, d), a.reps)[0]
            except Exception as e:
                continue
            res.append((us, tile, sk))
    auto = [r for r in res if r[1] == 0][0][0]
    res.sort()
    print(f"{name:22s} direct 3x3 {t_direct:7.1f} us | GEMM planner {auto:7.1f} | " + " ".join(f"{tl}/{s}:{us:.1f}" for us, tl, s in res[:8]), flush=True)
