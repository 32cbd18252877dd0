#!/usr/bin/env python3
"""Where does the wrong value first exist?  Runs the failing (tile, split-K 2) launch on the diagnostic twin built with -DMFC2_HZ=256
(medfusion_amd/csrc/build/variants/libmedfusion_hip_pk_dump.so): the surviving workgroup of every tile dumps its accumulator registers
right behind the split-K tree, before the epilogue touches them.  For every wrong OUTPUT element the script looks at the same element of
the dump: already wrong there (the tree lost it) or still right (the epilogue lost it)."""
import ctypes
import os
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch

from medfusion_amd import kernels as K
from medfusion_amd import lib as L

dev = torch.device("cuda:0")
TILES = {54: (128, 64, 2, 2), 53: (64, 128, 2, 2), 36: (128, 64, 4, 2)}
tile = int(sys.argv[1]) if len(sys.argv) > 1 else 54
BM, BN, WM, WN = TILES[tile]
FM, FN = BM // WM, BN // WN
TM, TN, NW = FM // 32, FN // 32, WM * WN
n, h, w, c1, co, k = 16, 16, 16, 512, 512, 3
M = n * h * w
g = torch.Generator().manual_seed(3)
wt = (torch.randn((co, k, k, c1), generator=g) * 0.02).to(dev)
b = torch.randn((co,), generator=g).to(dev)
x = torch.randn((n, h, w, c1), generator=g).to(dev)
wh = K.split_weight_f16x2(wt)
d2 = K.make_conv_desc(n, h, w, c1, 0, co, k, 1, 1, 0, tile_hint=tile, splitk_hint=2, precision=5)
dref = K.make_conv_desc(n, h, w, c1, 0, co, k, 1, 1, 0, tile_hint=31, splitk_hint=1, precision=5)
wa, wb = wt.clone(), wt.clone()
wa[..., c1 // 2:] = 0
wb[..., : c1 // 2] = 0
A = K.conv2d_f16x2(x, K.split_weight_f16x2(wa), None, dref).clone()
B = K.conv2d_f16x2(x, K.split_weight_f16x2(wb), None, dref).clone()
AB = A + B

lib = L.load()
setter = lib.mf_debug_set_conv_dump
setter.argtypes = [ctypes.c_void_p]
tiles_m, tiles_n = M // BM, co // BN
dump3 = torch.full((3, tiles_n * tiles_m, NW, TM, TN, 16, 64), float("nan"), device=dev)   # [sum behind the tree, own before the add, loaded]
dump = dump3[0]
setter(dump3.data_ptr())

# element (tile, wave, i, j, r, lane) -> (pixel m, channel c)   (conv_f16x2.h: epilogue comment)
t = torch.arange(tiles_n * tiles_m, device=dev).view(-1, 1, 1, 1, 1, 1)
wv = torch.arange(NW, device=dev).view(1, -1, 1, 1, 1, 1)
ii = torch.arange(TM, device=dev).view(1, 1, -1, 1, 1, 1)
jj = torch.arange(TN, device=dev).view(1, 1, 1, -1, 1, 1)
rr = torch.arange(16, device=dev).view(1, 1, 1, 1, -1, 1)
ln = torch.arange(64, device=dev).view(1, 1, 1, 1, 1, -1)
tile_m, tile_n = t % tiles_m, t // tiles_m
wm, wn = wv // WN, wv % WN
m = tile_m * BM + wm * FM + ii * 32 + (ln & 31)
c = tile_n * BN + wn * FN + jj * 32 + (rr & 3) + 8 * (rr >> 2) + 4 * (ln >> 5)
m, c = m.expand(dump.shape), c.expand(dump.shape)

for rep in range(200):
    dump3.fill_(float("nan"))
    y = K.conv2d_f16x2(x, wh, b, d2)
    bad_out = (y != AB + b)
    if not bool(bad_out.any()):
        continue
    want = AB.view(M, co)[m, c]
    bad_reg = dump != want
    print(f"launch {rep}: {int(bad_out.sum())} wrong OUTPUT elements; {int(bad_reg.sum())} wrong ACCUMULATOR elements in the dump taken right behind the tree "
          f"({int(torch.isnan(dump).sum())} never written)")
    out_at = bad_out.view(M, co)[m, c]
    both = int((out_at & bad_reg).sum())
    print(f"   wrong in the output AND already wrong in the dump: {both};  wrong in the output but RIGHT in the dump: {int((out_at & ~bad_reg).sum())};  "
          f"wrong in the dump but right in the output: {int((~out_at & bad_reg).sum())}")
    idx = (out_at | bad_reg).nonzero()
    lanes = sorted(set(idx[:, 5].tolist()))
    regs = sorted(set(idx[:, 4].tolist()))
    print(f"   lanes {lanes[0]}..{lanes[-1]} ({len(lanes)} distinct), accumulator registers {regs}")
    i0 = tuple(idx[0].tolist())
    mm, cc = int(m[i0]), int(c[i0])
    print(f"   e.g. tile {i0[0]} wave {i0[1]} block ({i0[2]},{i0[3]}) register {i0[4]} lane {i0[5]}: dump {float(dump[i0]):.7g}, y - bias {float(y.view(M, co)[mm, cc] - b[cc]):.7g}; "
          f"A {float(A.view(M, co)[mm, cc]):.7g}, B {float(B.view(M, co)[mm, cc]):.7g}, A + B {float(AB.view(M, co)[mm, cc]):.7g}")
    if not bool(torch.isnan(dump3[1]).all()):   # the twin built with -DMFC2_HZ=264 also dumps both operands of the add
        own, got = dump3[1], dump3[2]
        a_at, b_at = A.view(M, co)[m, c], B.view(M, co)[m, c]
        sel = bad_reg
        own_ok = ((own == a_at) | (own == b_at))[sel]
        got_ok = ((got == a_at) | (got == b_at))[sel]
        print(f"   at the {int(sel.sum())} wrong elements: own value is a correct K-slice partial in {int(own_ok.sum())}, loaded value is a correct partial in {int(got_ok.sum())}, "
              f"sum register == own in {int((dump == own)[sel].sum())}, == loaded in {int((dump == got)[sel].sum())}, == own + loaded in {int((dump == own + got)[sel].sum())}")
        print(f"   e.g. own {float(own[i0]):.7g}, loaded {float(got[i0]):.7g}, sum register {float(dump[i0]):.7g}")
        allel = ~torch.isnan(own)
        print(f"   over ALL elements: own correct partial {int(((own == a_at) | (own == b_at))[allel].sum())} / {int(allel.sum())}, loaded correct partial "
              f"{int(((got == a_at) | (got == b_at))[allel].sum())} / {int(allel.sum())}")
    break
else:
    print("no bad launch in 200 (the dump hook itself moved the code: see scripts/pk_hunt.py)")
