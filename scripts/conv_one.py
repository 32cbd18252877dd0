#!/usr/bin/env python3
"""Launch ONE convolution shape repeatedly (for rocprofv3 --pmc passes / quick A-B timing)."""
import argparse
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch

from medfusion_amd import kernels as K
from medfusion_amd import lib as L
import os
if os.environ.get("MF_LIB_OVERRIDE"):  # timing experiments with side builds (scripts/ablate_split.sh)
    L.LIB_PATH = Path(os.environ["MF_LIB_OVERRIDE"]).resolve()

ap = argparse.ArgumentParser()
ap.add_argument("--shape", default="16,32,32,256,0,256,3,1,0", help="N,H,W,C1,C2,Cout,k,stride,ups")
ap.add_argument("--tile", type=int, default=0)
ap.add_argument("--splitk", type=int, default=0)
ap.add_argument("--reps", type=int, default=20)
ap.add_argument("--precision", type=int, default=0)
ap.add_argument("--zeros", action="store_true", help="all-zero operands: the DVFS give-back of the same instruction stream (power-bound or not?)")
a = ap.parse_args()
n, h, w, c1, c2, co, k, st, ups = (int(v) for v in a.shape.split(","))
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
x1 = torch.randn((n, h, w, c1), generator=g).to(dev)
x2 = torch.randn((n, h, w, c2), generator=g).to(dev) if c2 else None
wt = (torch.randn((4, co, 2, 2, c1 + c2) if ups == 2 else (co, k, k, c1 + c2), generator=g) * 0.02).to(dev)
b = torch.randn((co,), generator=g).to(dev)
if a.zeros:
    x1.zero_(); wt.zero_()
    if x2 is not None:
        x2.zero_()
d = K.make_conv_desc(n, h, w, c1, c2, co, k, st, 1 if k == 3 else 0, ups, tile_hint=a.tile, splitk_hint=a.splitk, precision=a.precision)
if a.precision == 3:
    wt = K.split_conv_weight(wt)
if a.precision == 5:
    wt = K.split_weight_f16x2(wt)
    conv = K.conv2d_f16x2
else:
    conv = K.conv2d
y = conv(x1, wt, b, d, x2=x2)
for _ in range(5):
    conv(x1, wt, b, d, x2=x2, out=y)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(a.reps):
    conv(x1, wt, b, d, x2=x2, out=y)
e1.record()
torch.cuda.synchronize()
ho, wo = K.conv_out_hw(d)
gf = 2.0 * n * ho * wo * co * k * k * (c1 + c2) / 1e9
ms = e0.elapsed_time(e1) / a.reps
print(f"shape {a.shape} tile {a.tile} splitk {a.splitk}: {ms:.4f} ms  {gf / ms:.1f} TF")
