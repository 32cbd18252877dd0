#!/usr/bin/env python3
"""Determinism of the in-launch split-K reduction: the same convolution launched many times must give the same bits."""
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch

from medfusion_amd import kernels as K

dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(3)
bad_total = 0
for (n, h, w, c1, c2, co, k, tile, sk) in [(16, 8, 8, 1024, 1024, 1024, 3, 31, 8), (16, 16, 16, 512, 0, 512, 3, 33, 2), (16, 8, 8, 1024, 0, 1024, 3, 33, 4),
                                            (16, 16, 16, 1024, 0, 512, 3, 31, 4), (16, 8, 8, 1536, 0, 512, 3, 33, 8)]:
    x = torch.randn((n, h, w, c1), generator=g).to(dev)
    x2 = torch.randn((n, h, w, c2), generator=g).to(dev) if c2 else None
    wt = (torch.randn((co, k, k, c1 + c2), generator=g) * 0.02).to(dev)
    b = torch.randn((co,), generator=g).to(dev)
    wh = K.split_weight_f16x2(wt)
    d = K.make_conv_desc(n, h, w, c1, c2, co, k, 1, 1, 0, tile_hint=tile, splitk_hint=sk, precision=5)
    ref = K.conv2d_f16x2(x, wh, b, d, x2=x2).clone()
    d1 = K.make_conv_desc(n, h, w, c1, c2, co, k, 1, 1, 0, tile_hint=tile, splitk_hint=3, precision=5)   # slabs + reducer
    alt = K.conv2d_f16x2(x, wh, b, d1, x2=x2)
    bad = 0
    worst = 0.0
    parts = K.conv_gn_parts(d, 32)
    refp = None
    for rep in range(300):
        mode = rep % 3
        if mode == 0:
            y = K.conv2d_f16x2(x, wh, b, d, x2=x2)
        elif mode == 1:
            y = K.conv2d_f16x2(x, wh, b, d, x2=x2, measure_out=True)
            K.bound_of(y)
        else:
            y, part = K.conv2d_f16x2(x, wh, b, d, x2=x2, gn_groups=32, gn_parts=parts)
            if refp is None:
                refp = part.clone()
            elif not torch.equal(part, refp):
                bad += 1
        if not torch.equal(y, ref):
            bad += 1
            worst = max(worst, float((y - ref).abs().max()))
    print(f"shape {(n, h, w, c1, c2, co)} tile {tile} sk {sk}: {bad} of 300 launches differ (max abs diff {worst:.3e}); vs the reducer path rel {float((ref - alt).abs().max() / alt.abs().max()):.2e}", flush=True)
    bad_total += bad
print("TOTAL differing launches:", bad_total)
