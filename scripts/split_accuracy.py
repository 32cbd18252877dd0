#!/usr/bin/env python3
"""Error of the convolution arithmetic modes against an fp64 reference (run on the GPU box):
0 = fp32 MFMA, 3 = exact 3 x bf16 split (weights split at load), 5 = pairs of fp16 (23-bit operands, 3 product terms; LDS-DMA kernel)."""
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch
import torch.nn.functional as F

from medfusion_amd import kernels as K

dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
print(f"{'shape':34s} {'mode0':>9s} {'mode3':>9s} {'mode5':>9s}   (max |y - y64| / max |y64|;  rms)")
for n, h, c1, co, k in ((2, 8, 32, 64, 3), (2, 16, 256, 256, 3), (2, 8, 1024, 1024, 3), (2, 8, 2048, 1024, 3), (2, 16, 512, 256, 1), (4, 32, 256, 256, 3)):
    x = torch.randn((n, c1, h, h), generator=g)
    w = torch.randn((co, c1, k, k), generator=g) / (c1 * k * k) ** 0.5
    b = torch.randn((co,), generator=g) * 0.1
    want = F.conv2d(x.to(dev).double(), w.to(dev).double(), b.to(dev).double(), padding=k // 2)
    xd, wp, bd = K.nchw_to_nhwc(x.to(dev)), K.pack_conv_weight(w.to(dev)), b.to(dev)
    errs = []
    for mode in (0, 3, 5):
        d = K.make_conv_desc(n, h, h, c1, 0, co, k, 1, k // 2, 0, precision=mode)
        y = K.nhwc_to_nchw(K.conv2d_f16x2(xd, K.split_weight_f16x2(wp), bd, d) if mode == 5 else
                           K.conv2d(xd, K.split_conv_weight(wp) if mode == 3 else wp, bd, d)).double()
        errs.append((float((y - want).abs().max() / want.abs().max()), float(((y - want) ** 2).mean().sqrt() / (want ** 2).mean().sqrt())))
    print(f"{str((n, h, c1, co, k)):34s} " + " ".join(f"{e[0]:9.2e}" for e in errs) + "   rms " + " ".join(f"{e[1]:8.2e}" for e in errs))
    if c1 * k * k >= 9216:   # mode 5 against the length of one accumulation chain (chunks of 32): does the matrix core's truncating add drift?
        wh = K.split_weight_f16x2(wp)
        line = []
        for sk in (1, 2, 4, 8):
            d = K.make_conv_desc(n, h, h, c1, 0, co, k, 1, k // 2, 0, tile_hint=33, splitk_hint=sk, precision=5)
            y = K.nhwc_to_nchw(K.conv2d_f16x2(xd, wh, bd, d)).double()
            line.append(f"chain {c1 * k * k // 32 // sk}: {float((y - want).abs().max() / want.abs().max()):.2e}")
        print("    mode 5, " + "  ".join(line))
