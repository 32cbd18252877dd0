#!/usr/bin/env python3
"""What one launch of the fp16-pair convolution costs besides its K loop: the same output tile grid (N, H, W, Cout fixed) timed over
input widths 32 .. 512, i.e. 9 .. 144 chunks per workgroup, and fitted as  t = fixed + per_chunk x chunks.  Device time (command list
replay).  Run on the GPU box."""
import argparse
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch

from medfusion_amd import kernels as K
from _devtime import device_us


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--tiles", default="33,53,52")
    ap.add_argument("--geoms", default="16,32,256;16,16,512;16,8,1024", help="N,H(=W),Cout;...")
    ap.add_argument("--k", type=int, default=3)
    ap.add_argument("--gn", type=int, default=32, help="GroupNorm groups whose partial records the epilogue leaves (0: none)")
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    g = torch.Generator(device="cpu").manual_seed(0)
    for geom in args.geoms.split(";"):
        n, h, co = (int(v) for v in geom.split(","))
        for tile in (int(t) for t in args.tiles.split(",")):
            pts = []
            for cin in (32, 64, 128, 256, 512):
                x = torch.randn((n, h, h, cin), generator=g).to(dev)
                w = (torch.randn((co, args.k, args.k, cin), generator=g) * 0.02).to(dev)
                b = torch.randn((co,), generator=g).to(dev)
                d = K.make_conv_desc(n, h, h, cin, 0, co, args.k, 1, 1 if args.k == 3 else 0, 0, tile_hint=tile, splitk_hint=1, precision=5)
                if not K.conv_f16x2_ok(d):
                    continue
                wh = K.split_weight_f16x2(w)
                parts = K.conv_gn_parts(d, args.gn) if args.gn else 0
                y = torch.empty((n, h, h, co), device=dev)

                def run():
                    if parts:
                        return K.conv2d_f16x2(x, wh, b, d, out=y, gn_groups=args.gn, gn_parts=parts)
                    return K.conv2d_f16x2(x, wh, b, d, out=y)
                us, nl = device_us(run)
                pts.append((cin // 32 * args.k * args.k, us, nl))
            if len(pts) < 3:
                continue
            # least squares over the three longest
            xs, ys = [p[0] for p in pts[-3:]], [p[1] for p in pts[-3:]]
            mx, my = sum(xs) / 3, sum(ys) / 3
            slope = sum((a - mx) * (b_ - my) for a, b_ in zip(xs, ys)) / sum((a - mx) ** 2 for a in xs)
            print(f"N={n} {h}x{h} Cout={co} k={args.k} tile {tile} gn={args.gn}: " + "  ".join(f"{c}ch:{u:.1f}us" for c, u, _ in pts) +
                  f"  | launches/call {pts[-1][2]}  fit: fixed {my - slope * mx:.1f} us + {slope * 1e3:.0f} ns/chunk", flush=True)


if __name__ == "__main__":
    main()
