#!/usr/bin/env python3
"""Per-shape timing of every convolution of the published UNet (B per GPU) and VAE decoder, over the implicit-GEMM
tile configs and split-K factors (tuning tool for the planner in csrc/conv.hip; run on the GPU box)."""
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


def unet_shapes(B):
    # (name, N, H, W, C1, C2, Cout, k, stride, ups, count per UNet call)
    S = []
    def res(name, h, c1, c2, co, n=1):
        S.append((f"{name}.c0", B, h, h, c1, c2, co, 3, 1, 0, n))
        S.append((f"{name}.c1", B, h, h, co, 0, co, 3, 1, 0, n))
        if c1 + c2 != co:
            S.append((f"{name}.res", B, h, h, c1, c2, co, 1, 1, 0, n))
    res("in32 R256", 32, 256, 0, 256, 2)
    S.append(("down32", B, 32, 32, 256, 0, 256, 3, 2, 0, 1))
    res("in16 R256-512", 16, 256, 0, 512)
    res("in16 R512", 16, 512, 0, 512)
    S.append(("down16", B, 16, 16, 512, 0, 512, 3, 2, 0, 1))
    res("in8 R512-1024", 8, 512, 0, 1024)
    res("in8/mid R1024", 8, 1024, 0, 1024, 3)
    res("out8 R2048-1024", 8, 1024, 1024, 1024, 2)
    res("out8 R1536-512", 8, 1024, 512, 512)
    S.append(("up8->16", B, 8, 8, 512, 0, 512, 3, 1, 1, 1))
    res("out16 R1024-512", 16, 512, 512, 512, 2)
    res("out16 R768-256", 16, 512, 256, 256)
    S.append(("up16->32", B, 16, 16, 256, 0, 256, 3, 1, 1, 1))
    res("out32 R512-256", 32, 256, 256, 256, 3)
    return S


def vae_shapes(B):
    S = []
    S.append(("vae R512@32.c1", B, 32, 32, 512, 0, 512, 3, 1, 0, 1))
    for h, ci, co in ((32, 512, 256), (64, 256, 128), (128, 128, 64)):
        S.append((f"vae up{h}", B, h, h, ci, 0, co, 3, 1, 1, 1))
        S.append((f"vae R{co}@{2*h}", B, 2 * h, 2 * h, co, 0, co, 3, 1, 0, 2))
    return S


_W3 = {}


def _split_cache(w):
    k = w.data_ptr()
    if k not in _W3:
        _W3.clear()
        _W3[k] = K.split_conv_weight(w)
    return _W3[k]


_WH = {}


def time_conv(x1, x2, w, b, d, reps):
    if d.precision in (5, 6):
        k = w.data_ptr()
        if k not in _WH:
            _WH.clear()
            _WH[k] = K.split_weight_f16x2(w)
        wh = _WH[k]
        y = K.conv2d_f16x2(x1, wh, b, d, x2=x2)   # (the activations' fp16-pair mirrors are cached on the tensors by this first call)
        torch.cuda.synchronize()
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
        ev[0].record()
        for _ in range(reps):
            K.conv2d_f16x2(x1, wh, b, d, x2=x2, out=y)
        ev[1].record()
        torch.cuda.synchronize()
        return ev[0].elapsed_time(ev[1]) / reps
    if d.precision == 3:
        w = _split_cache(w)
    y = K.conv2d(x1, w, b, d, x2=x2)
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    ev[0].record()
    for _ in range(reps):
        K.conv2d(x1, w, b, d, x2=x2, out=y)
    ev[1].record()
    torch.cuda.synchronize()
    return ev[0].elapsed_time(ev[1]) / reps


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=16)
    ap.add_argument("--vae-batch", type=int, default=4)
    ap.add_argument("--reps", type=int, default=8)
    ap.add_argument("--quick", action="store_true", help="auto config only")
    ap.add_argument("--only", default="", help="substring filter on the shape name")
    ap.add_argument("--precision", type=int, default=0, help="0: fp32 MFMA, 3: fp32 split exactly into 3 bf16 terms (weights split at load), 4: bf16 (opt-in), 5: fp16 pairs (LDS-DMA kernel), 6: single-term fp16 on the same kernel (opt-in)")
    ap.add_argument("--tiles", default="", help="comma list of tile ids to sweep (default: all built for the precision)")
    ap.add_argument("--latent", type=int, default=32, help="UNet input size (32: 256-px models, 64: 512-px)")
    ap.add_argument("--emit-table", default="", help="append the best (tile, split-K) of every shape to this file as planner table entries (precision 5)")
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    g = torch.Generator(device="cpu").manual_seed(0)
    tot_auto = tot_best = tot_fl = 0.0
    print(f"{'shape':24s} {'M':>7s} {'N':>5s} {'K':>6s} {'GF':>7s} | {'auto ms':>8s} {'TF':>6s} | best cfg (tile,splitk) ms TF | all")
    table = []
    shapes = unet_shapes(args.batch) + vae_shapes(args.vae_batch)
    if args.latent != 32:   # same architecture on a larger latent: every spatial size scales
        f = args.latent // 32
        shapes = [(nm, n, h * f, w * f, c1, c2, co, k, st, ups, cnt) for nm, n, h, w, c1, c2, co, k, st, ups, cnt in shapes]
    for name, n, h, w_, c1, c2, co, k, st, ups, cnt in shapes:
        if args.only and args.only not in name:
            continue
        pad = 1 if k == 3 else 0
        if args.precision in (5, 6) and ups:
            ups = 2
        x1 = torch.randn((n, h, w_, c1), generator=g).to(dev)
        x2 = torch.randn((n, h, w_, c2), generator=g).to(dev) if c2 else None
        wt = (torch.randn((4, co, 2, 2, c1 + c2) if ups == 2 else (co, k, k, c1 + c2), generator=g) * 0.02).to(dev)
        b = torch.randn((co,), generator=g).to(dev)
        d0 = K.make_conv_desc(n, h, w_, c1, c2, co, k, st, pad, ups, precision=args.precision)
        ho, wo = K.conv_out_hw(d0)
        M, Kk = n * ho * wo, k * k * (c1 + c2)
        gf = 2.0 * M * co * Kk / 1e9
        if args.precision in (5, 6) and not K.conv_f16x2_ok(d0):
            print(f"{name:24s} not on the fp16-pair kernel")
            continue
        time_conv(x1, x2, wt, b, d0, args.reps)  # clock/cache warm-up
        t_auto = time_conv(x1, x2, wt, b, d0, args.reps)
        res = []
        if not args.quick:
            tiles = [int(t) for t in args.tiles.split(",")] if args.tiles else ((31, 32, 33, 34, 35, 36, 37, 51, 52, 53, 54, 61, 62, 63, 64) if args.precision in (5, 6) else (1, 3, 4, 7, 8, 9, 10) if args.precision else (1, 3, 4, 7, 8, 9, 23, 24, 27, 28))
            for tile in tiles:
                bn = {61: 128, 62: 128, 63: 128, 64: 128, 31: 256, 32: 128, 33: 128, 34: 128, 35: 64, 36: 64, 37: 256, 51: 128, 52: 128, 53: 128, 54: 64, 1: 128, 2: 64, 3: 128, 4: 64, 7: 128, 8: 128, 9: 256, 10: 128, 11: 128, 12: 128, 13: 64, 23: 128, 24: 64, 27: 128, 28: 128}[tile]
                bm = {61: 256, 62: 256, 63: 128, 64: 128, 31: 128, 32: 256, 33: 128, 34: 128, 35: 256, 36: 128, 37: 64, 51: 128, 52: 128, 53: 64, 54: 128, 1: 128, 2: 128, 3: 64, 4: 64, 7: 128, 8: 128, 9: 128, 10: 256, 11: 128, 12: 64, 13: 128, 23: 64, 24: 64, 27: 128, 28: 128}[tile]
                if ups == 2 and (h * w_) % bm:
                    continue
                if tile in (23, 24, 27, 28) and (c1 % 64 or c2 % 64):
                    continue
                if co % bn:
                    continue
                for sk in (1, 2, 4, 8, 16):
                    if sk > 1 and Kk // 32 // sk < 8:
                        continue
                    if args.precision in (5, 6) and sk > (c1 + c2) // 32:
                        continue
                    if args.precision in (5, 6) and -(-((c1 + c2) // 32) // sk) * (4 if ups == 2 else k * k) > 96:
                        continue   # one accumulation chain <= 96 chunks (profiles/r02_split_accuracy.txt)
                    d = K.make_conv_desc(n, h, w_, c1, c2, co, k, st, pad, ups, tile_hint=tile, splitk_hint=sk, precision=args.precision)
                    if args.precision in (5, 6) and not K.conv_f16x2_ok(d):
                        continue   # (a halo tile this geometry does not fit)
                    res.append((time_conv(x1, x2, wt, b, d, args.reps), tile, sk))
            res.sort()
            t_auto = min(t_auto, time_conv(x1, x2, wt, b, d0, args.reps))   # (the first timing of a shape runs on clocks that are still ramping: long VAE rows lose 20 %)
        best = res[0] if res else (t_auto, 0, 0)
        if res and best[0] < 0.985 * t_auto:   # keep the planner's own choice unless the sweep beats it by more than the noise
            table.append((n, h, w_, c1 + c2, co, k, st, ups, best[1], best[2]))
        tot_auto += t_auto * cnt
        tot_best += best[0] * cnt
        tot_fl += gf * cnt
        allres = " ".join(f"{t}/{s}:{ms:.3f}" for ms, t, s in res[:8])
        print(f"{name:24s} {M:7d} {co:5d} {Kk:6d} {gf:7.2f} | {t_auto:8.3f} {gf / t_auto:6.1f} | ({best[1]},{best[2]}) {best[0]:.3f} {gf / best[0]:6.1f} | {allres}", flush=True)
    if args.emit_table and table:
        with open(args.emit_table, "a") as f:
            for e in table:
                f.write("    {%s},\n" % ", ".join(str(v) for v in e))
    print(f"TOTAL weighted: auto {tot_auto:.2f} ms ({tot_fl / tot_auto:.1f} TF)  best {tot_best:.2f} ms ({tot_fl / tot_best:.1f} TF)  flops {tot_fl:.1f} GF")


if __name__ == "__main__":
    main()
