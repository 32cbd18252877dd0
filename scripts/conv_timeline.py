#!/usr/bin/env python3
"""Where the fixed cost of a convolution launch goes: the diagnostic twin built with -DMFC2_HZ=512 (medfusion_amd.build.build_variant("stamp",
conv_flags=["-DMFC2_HZ=512"]); run with MEDFUSION_LIB=<that library>) makes every workgroup leave the 100 MHz real-time counter at its entry,
when its first chunk has landed, at the end of its K loop and at the end of its epilogue.  Two back-to-back launches of the same convolution
into two stamp buffers give, in one clock: the boundary between the launches, the dispatch skew, the ramp, the loop, the drain, the tail skew."""
import argparse
import ctypes
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch

from medfusion_amd import kernels as K
from medfusion_amd import lib as L


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--fused", action="store_true", help="time mf_conv2d_f16x2_gn_apply (GroupNorm + Swish + residual + embedding inside the launch) instead")
    ap.add_argument("--shapes", default="16,32,32,256,0,256,3:52:1;16,16,16,512,0,512,3:52:2;16,8,8,1024,0,1024,3:53:4;16,32,32,512,0,256,1:52:1")
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    lib = L.load()
    setter = lib.mf_debug_set_conv_dump
    setter.argtypes = [ctypes.c_void_p]
    g = torch.Generator().manual_seed(0)
    print("times in us (10 ns counter).  boundary = first entry of launch 2 - last exit of launch 1; entry skew = last entry - first entry; "
          "ramp = entry -> first chunk landed; drain = end of the K loop -> exit (epilogue); tail = last exit - median exit")
    for spec in args.shapes.split(";"):
        shp, tile, sk = spec.split(":")
        n, h, w, c1, c2, co, k = (int(v) for v in shp.split(","))
        x = torch.randn((n, h, w, c1), generator=g).to(dev)
        wt = (torch.randn((co, k, k, c1 + c2), generator=g) * 0.02).to(dev)
        b = torch.randn((co,), generator=g).to(dev)
        d = K.make_conv_desc(n, h, w, c1, c2, co, k, 1, 1 if k == 3 else 0, 0, tile_hint=int(tile), splitk_hint=int(sk), precision=5)
        wh = K.split_weight_f16x2(wt)
        parts = K.conv_gn_parts(d, 32)
        y = torch.empty((n, h, w, co), device=dev)
        setter(None)
        if args.fused:
            words = K.conv_fuse_words(d, 32)
            gamma, beta = torch.rand((co,), device=dev) + 0.5, torch.randn((co,), device=dev) * 0.1
            res = torch.randn((n, h, w, co), device=dev)
            K.split_of(res)
            emb = torch.randn((n, co), device=dev)
            emb._mf_bound = emb.abs().amax(1)
            if not words:
                print(f"{shp} tile {tile} split-K {sk}: no fused form")
                continue
            run = lambda: K.conv2d_f16x2_gn_apply(x, wh, b, d, gamma, beta, 32, 1e-5, parts, words, act=1, residual=res, emb=emb, emb_stride=co, bconst=30.0, out_fp32=False)
        else:
            run = lambda: K.conv2d_f16x2(x, wh, b, d, out=y, gn_groups=32, gn_parts=parts)
        for _ in range(5):
            run()
        bufs = [torch.zeros((8192, 8), dtype=torch.int64, device=dev) for _ in range(3)]
        torch.cuda.synchronize()
        for bf in bufs:
            setter(bf.data_ptr())
            run()
        setter(None)
        torch.cuda.synchronize()
        A, B = (bf[bf[:, 0] > 0].double() * 0.01 for bf in bufs[1:])   # us
        grid = B.shape[0]
        t0 = B[:, 0].min()
        Bx = B[B[:, 3] > 0]      # (a workgroup that hands its split-K partial to its partner leaves from inside the tree: no exit stamp)
        med = lambda v: float(v.median())
        print(f"{shp} tile {tile} split-K {sk}: {grid} workgroups | boundary {float(t0 - A[:, 3].max()):5.2f} | entry skew {float(B[:, 0].max() - t0):5.2f} | "
              f"ramp median {med(B[:, 1] - B[:, 0]):5.2f} max {float((B[:, 1] - B[:, 0]).max()):5.2f} | loop median {med(B[:, 2] - B[:, 1]):6.2f} | "
              f"drain median {med(Bx[:, 3] - Bx[:, 2]):5.2f} max {float((Bx[:, 3] - Bx[:, 2]).max()):5.2f} | tail {float(Bx[:, 3].max()) - med(Bx[:, 3]):5.2f} | "
              f"first entry -> last exit {float(B[:, 3].max() - t0):6.2f} || inside the drain (medians): wait for the other waves {med(Bx[:, 4] - Bx[:, 2]):5.2f}, "
              f"split-K tree {med(Bx[:, 5] - Bx[:, 4]):5.2f}, " + (f"staging + records + arrival {med(Bx[:, 6] - Bx[:, 5]):5.2f}, wait for the sample (constants and residual loads issued) "
              f"median {med(Bx[:, 7] - Bx[:, 6]):5.2f} max {float((Bx[:, 7] - Bx[:, 6]).max()):5.2f} min {float((Bx[:, 7] - Bx[:, 6]).min()):5.2f}, finalize + apply + stores {med(Bx[:, 3] - Bx[:, 7]):5.2f}"
              if args.fused else f"staging + stores {med(Bx[:, 6] - Bx[:, 5]):5.2f}, bounds + GroupNorm records {med(Bx[:, 3] - Bx[:, 6]):5.2f}"), flush=True)


if __name__ == "__main__":
    main()
