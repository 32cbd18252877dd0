#!/usr/bin/env python3
"""The in-launch split-K reduction of mf_conv2d_f16x2 under the conditions that expose a stale hand-off: the launches ALTERNATE between
different inputs (a workgroup that read its partner's slot too early, or through a stale cache line, would mix in the previous launch's
tile), every result is compared with the un-split launch of the same input, and equal inputs must give equal bits."""
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch

from medfusion_amd import kernels as K

dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(3)
total_bad = 0
CASES = [(16, 32, 32, 256, 256, 256, 3, 32, 2), (16, 8, 8, 1024, 1024, 1024, 3, 31, 8), (16, 16, 16, 512, 0, 512, 3, 33, 2),
         (16, 8, 8, 1024, 0, 1024, 3, 33, 4), (16, 16, 16, 1024, 0, 512, 3, 51, 4), (16, 8, 8, 1536, 0, 512, 3, 53, 8),
         (16, 16, 16, 512, 512, 512, 1, 53, 2), (16, 16, 16, 512, 0, 512, 3, 53, 2), (16, 16, 16, 512, 0, 512, 3, 54, 2), (16, 16, 16, 512, 0, 512, 3, 36, 2),
         (16, 16, 16, 512, 0, 512, 3, 52, 2), (16, 32, 32, 256, 0, 256, 3, 53, 2), (16, 16, 16, 512, 0, 512, 3, 53, 3)]
if len(sys.argv) > 1:
    CASES = [c for c in CASES if c[7] in [int(v) for v in sys.argv[1].split(",")]]
for (n, h, w, c1, c2, co, k, tile, sk) in CASES:
    pad = 1 if k == 3 else 0
    wt = (torch.randn((co, k, k, c1 + c2), generator=g) * 0.02).to(dev)
    b = torch.randn((co,), generator=g).to(dev)
    wh = K.split_weight_f16x2(wt)
    d = K.make_conv_desc(n, h, w, c1, c2, co, k, 1, pad, 0, tile_hint=tile, splitk_hint=sk, precision=5)
    d1 = K.make_conv_desc(n, h, w, c1, c2, co, k, 1, pad, 0, tile_hint=tile, splitk_hint=1, precision=5)
    xs = [(torch.randn((n, h, w, c1), generator=g).to(dev) * (1 + 3 * i), torch.randn((n, h, w, c2), generator=g).to(dev) if c2 else None) for i in range(3)]
    refs = [K.conv2d_f16x2(x, wh, b, d1, x2=x2).clone() for x, x2 in xs]
    firsts = [None] * 3
    bad = 0
    worst = 0.0
    for rep in range(90):
        i = (rep * 7 + rep // 5) % 3
        x, x2 = xs[i]
        y = K.conv2d_f16x2(x, wh, b, d, x2=x2, measure_out=bool(rep % 2))
        err = float((y - refs[i]).abs().max() / refs[i].abs().max())
        worst = max(worst, err)
        if err > 1e-5:
            bad += 1
        if firsts[i] is None:
            firsts[i] = y.clone()
        elif not torch.equal(y, firsts[i]):
            bad += 1
    print(f"shape {(n, h, w, c1, c2, co, k)} tile {tile} sk {sk}: {bad} bad of 90 alternating launches (worst rel err vs the un-split launch {worst:.2e})", flush=True)
    total_bad += bad
print("TOTAL bad launches:", total_bad)
