#!/usr/bin/env python3
"""Upper bound on a fused-output Winograd F(2x2,3x3) kernel for the 32 x 32 level (VERDICT r05 Next #1), measured on the EXISTING kernel.

A workgroup of the fused form owns (BM tile rows x BN output channels) and loops over the 16 components itself: its K loop is
16 x Cin / 32 chunks of the same LDS-DMA / MFMA stream the component GEMM runs today, plus what this proxy leaves out (the fold of every
component's accumulator into four output accumulators, a four-times-larger epilogue, the exchange between workgroups when the
components are split over several of them).  The stream itself is exactly a 1x1 convolution with Cin' = 16 Cin over M = N (H/2)(W/2) rows:
  R256 @32^2, B = 16 :  M = 4096, Cout = 256, Cin' = 4096     (executed MACs = 4/9 of the direct 3x3)
  R512-256 @32^2     :  M = 4096, Cout = 256, Cin' = 8192
timed here per (tile, split-K).  split-K = 4 in-launch (the tree) stands for "four workgroups own four components each and meet inside the
launch" (its hand-off moves ONE BM x BN tile per partner; the real exchange moves two).  Everything the proxy omits costs time: a proxy that
does not beat the direct convolution by a wide margin settles the question without building the kernel.
"""
import argparse
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
sys.path.insert(0, str(Path(__file__).resolve().parent))
import torch

from medfusion_amd import kernels as K
from _devtime import device_us

TILES = {31: (128, 256), 32: (256, 128), 33: (128, 128), 34: (128, 128), 35: (256, 64), 36: (128, 64), 37: (64, 256), 51: (128, 128), 52: (128, 128),
         53: (64, 128), 54: (128, 64)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=16)
    ap.add_argument("--reps", type=int, default=20)
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(0)
    n = a.batch
    print(f"# B = {n}: the K loop of a fused-output Winograd workgroup as a 1x1 convolution over 16 Cin channels (device us, command-list replay)")
    for name, cin, direct_shape in (("R256@32", 256, (n, 32, 32, 256, 0, 256)), ("R512-256@32", 512, (n, 32, 32, 256, 256, 256))):
        # the direct 3x3 it would replace, planner's plan
        dn, dh, dw, c1, c2, co = direct_shape
        x1 = torch.randn((dn, dh, dw, c1), generator=g).to(dev)
        x2 = torch.randn((dn, dh, dw, c2), generator=g).to(dev) if c2 else None
        wd = K.split_weight_f16x2((torch.randn((co, 3, 3, c1 + c2), generator=g) * 0.02).to(dev))
        b = torch.randn((co,), generator=g).to(dev)
        dd = K.make_conv_desc(dn, dh, dw, c1, c2, co, 3, 1, 1, 0, precision=5)
        yd = K.conv2d_f16x2(x1, wd, b, dd, x2=x2)
        t_direct = device_us(lambda: K.conv2d_f16x2(x1, wd, b, dd, x2=x2, out=yd), a.reps)[0]
        print(f"{name}: direct 3x3 (plan {K.conv_plan(dd)}) {t_direct:.1f} us")
        # the proxy: M = n * 256 rows, 16 cin channels, 1x1
        kc = 16 * cin
        xp = torch.randn((n, 16, 16, kc), generator=g).to(dev)
        wp = K.split_weight_f16x2((torch.randn((co, 1, 1, kc), generator=g) * 0.02).to(dev))
        res = []
        for tile, (bm, bn) in TILES.items():
            if (n * 256) % bm or co % bn:
                continue
            for sk in (1, 2, 4, 8):
                d = K.make_conv_desc(n, 16, 16, kc, 0, co, 1, 1, 0, 0, tile_hint=tile, splitk_hint=sk, precision=5)
                if not K.conv_f16x2_ok(d):
                    continue
                try:
                    y = K.conv2d_f16x2(xp, wp, b, d)
                    t = device_us(lambda: K.conv2d_f16x2(xp, wp, b, d, out=y), a.reps)[0]
                except Exception as e:   # a plan the library refuses (reducer-pass split-K etc.)
                    print(f"   tile {tile} sk {sk}: {str(e)[:80]}")
                    continue
                wgs = (n * 256 // bm) * (co // bn) * sk
                res.append((t, tile, sk, wgs))
        res.sort()
        print("   proxy us by (tile/split-K [workgroups]): " + " ".join(f"{t}/{s}[{w}]:{us:.1f}" for us, t, s, w in res[:14]), flush=True)
        best = res[0]
        gexec = 2.0 * 3 * n * 256 * co * kc / 1e9
        print(f"   best proxy {best[0]:.1f} us = {gexec / best[0] * 1e3:.0f} TF executed ({gexec / best[0] * 1e3 / 2516.8:.2f} of nominal); direct {t_direct:.1f} us; "
              f"proxy / direct = {best[0] / t_direct:.2f}")


if __name__ == "__main__":
    main()
