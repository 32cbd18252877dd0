#!/usr/bin/env python3
"""What IS the wrong value?  (diagnostic for the packed-fp32 build of conv_f16x2_kernel; run with MEDFUSION_LIB=<the pk twin>)

For one failing (shape, tile, split-K 2) the script builds, with un-split launches that are exact by construction, every value the
workgroups of the split launch ever hold for an output element -- the two K-slice partials A and B (bias-free), their sum S -- and then,
for each wrong output y[p, c] of a bad launch, searches ALL positions (p', c') for one whose value explains it bit for bit:
    y == fl(X[p', c'] + bias[c])     X in {A, B, S, 0}
and prints where the matching value lives relative to the wrong element (same pixel?  channel distance?  the other K slice?).
"""
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch

from medfusion_amd import kernels as K

dev = torch.device("cuda:0")
n, h, w, c1, co, k = 16, 16, 16, 512, 512, 3
tile = int(sys.argv[1]) if len(sys.argv) > 1 else 54
g = torch.Generator().manual_seed(3)
wt = (torch.randn((co, k, k, c1), generator=g) * 0.02).to(dev)
b = torch.randn((co,), generator=g).to(dev)
x = torch.randn((n, h, w, c1), generator=g).to(dev)
wh = K.split_weight_f16x2(wt)
d2 = K.make_conv_desc(n, h, w, c1, 0, co, k, 1, 1, 0, tile_hint=tile, splitk_hint=2, precision=5)
dref = K.make_conv_desc(n, h, w, c1, 0, co, k, 1, 1, 0, tile_hint=31, splitk_hint=1, precision=5)


def unsplit(xin, win, bias):
    return K.conv2d_f16x2(xin, K.split_weight_f16x2(win) if win is not wt else wh, bias, dref).clone()


# K slice 0 = channels [0, 256), slice 1 = [256, 512) (conv_f16x2.h: a slice holds ALL taps of its 32-channel chunks).  The activations keep
# their per-sample scale (bound) when half the channels are zeroed only if the bound is unchanged: zero the WEIGHTS of the other half instead.
wa, wb = wt.clone(), wt.clone()
wa[..., c1 // 2:] = 0
wb[..., : c1 // 2] = 0
# (the weight scale of a split is a power of two: a different max only moves the exponent, the values stay exact)
A, B = unsplit(x, wa, None), unsplit(x, wb, None)
S = unsplit(x, wt, None)
ref = unsplit(x, wt, b)
print("A + B == S bit for bit:", bool(torch.equal(A + B, S)), "  (max |A+B-S| / max|S| = %.2e)" % float((A + B - S).abs().max() / S.abs().max()))
cands = {"A + B (the correct value of the tree)": A + B, "A (K slice 0 alone)": A, "B (K slice 1 alone)": B, "S (the un-split chain)": S,
         "0": torch.zeros_like(S), "2A": 2 * A, "2B": 2 * B}
first = None
need = K.pin_conv_plan(d2)[0]
poison = len(sys.argv) > 2 and sys.argv[2] == "poison"
for rep in range(400):
    if poison:   # the hand-off slots hold NaN patterns before the launch: a value read from a slot nobody wrote yet shows as NaN
        K.Workspace.get(need, dev).fill_(0xFF)
    y = K.conv2d_f16x2(x, wh, b, d2)
    if first is None:
        first = y.clone()
    bad = (y != (A + B) + b).nonzero()
    if bad.shape[0]:
        print(f"launch {rep}: {bad.shape[0]} wrong elements, {int(torch.isnan(y).sum())} of them NaN" + (" (slots poisoned with NaN before the launch)" if poison else ""))
        tally = {}
        for idx in bad[:64].tolist():
            nn, yy, xx, cc = idx
            val = y[nn, yy, xx, cc]
            found = []
            for name, X in cands.items():
                hit = ((X + b[cc]) == val).nonzero()
                for hnn, hyy, hxx, hcc in hit[:3].tolist():
                    dp = ((hnn * h + hyy) * w + hxx) - ((nn * h + yy) * w + xx)
                    found.append(f"{name} at pixel {dp:+d}, channel {hcc - cc:+d}")
            key = found[0] if found else "no exact match in A, B, S, 0 (+ bias)"
            tally[key] = tally.get(key, 0) + 1
        for key, cnt in sorted(tally.items(), key=lambda kv: -kv[1]):
            print(f"   {cnt:3d} x  {key}")
        i0 = tuple(bad[0].tolist())
        print(f"   e.g. {list(i0)}: y - bias = {float(y[i0] - b[i0[3]]):.7g}; A = {float(A[i0]):.7g}, B = {float(B[i0]):.7g}, S = {float(S[i0]):.7g}")
        break
else:
    print("no bad launch in 400")
