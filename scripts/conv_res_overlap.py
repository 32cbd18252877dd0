#!/usr/bin/env python3
"""Experiment: does the 1x1 conv_res of a channel-changing ResBlock hide behind its 3x3 sibling when the two run on
two streams?  Both read the block input and are independent (conv_blocks.py:238 vs :185); the sibling fills the chip with one 96 KB (or
72 KB) workgroup per CU, so the 1x1 only overlaps if its workgroups are CO-RESIDENT: a tile whose LDS fits beside the sibling's.
For every such pair of the published UNet at B = 16: time of the sibling alone, the 1x1 alone, both back to back on one stream, both on two
streams (fork / join by events, as a second stream inside the denoise loop would), over the 1x1's tile choices."""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch
from medfusion_amd import kernels as K

dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
# (name, N, H, W, C1, C2, Cout, sibling (tile, split-K))
PAIRS = [("in16 R256-512", 16, 16, 16, 256, 0, 512, (36, 1)), ("in8 R512-1024", 16, 8, 8, 512, 0, 1024, (53, 2)),
         ("out8 R2048-1024", 16, 8, 8, 1024, 1024, 1024, (51, 8)), ("out8 R1536-512", 16, 8, 8, 1024, 512, 512, (53, 8)),
         ("out16 R1024-512", 16, 16, 16, 512, 512, 512, (52, 4)), ("out16 R768-256", 16, 16, 16, 512, 256, 256, (52, 4)),
         ("out32 R512-256", 16, 32, 32, 256, 256, 256, (52, 2))]
RES_TILES = [(0, 0), (51, 1), (53, 1), (53, 2), (54, 1), (36, 1)]
REPS = 200
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()


def timed(fn):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    with torch.cuda.stream(s1):
        e0.record()
        for _ in range(REPS):
            fn()
        e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / REPS * 1e3


print(f"{'pair':18s} {'1x1 tile':>9s} | {'3x3 us':>7s} {'1x1 us':>7s} {'serial':>7s} {'2 streams':>9s} | hidden of the 1x1")
for name, n, h, w, c1, c2, co, (t3, k3) in PAIRS:
    x1 = torch.randn((n, h, w, c1), generator=g).to(dev)
    x2 = torch.randn((n, h, w, c2), generator=g).to(dev) if c2 else None
    w3 = K.split_weight_f16x2((torch.randn((co, 3, 3, c1 + c2), generator=g) * 0.02).to(dev))
    w1 = K.split_weight_f16x2((torch.randn((co, 1, 1, c1 + c2), generator=g) * 0.05).to(dev))
    b = torch.randn((co,), generator=g).to(dev)
    d3 = K.make_conv_desc(n, h, w, c1, c2, co, 3, 1, 1, 0, tile_hint=t3, splitk_hint=k3, precision=5)
    parts = K.conv_gn_parts(d3, 32)
    y3, y1 = torch.empty((n, h, w, co), device=dev), torch.empty((n, h, w, co), device=dev)
    K.split_of(x1)
    if x2 is not None:
        K.split_of(x2)
    for t1, k1 in RES_TILES:
        d1 = K.make_conv_desc(n, h, w, c1, c2, co, 1, 1, 0, 0, tile_hint=t1, splitk_hint=k1, precision=5)
        if not K.conv_f16x2_ok(d1):
            continue

        def f3():
            K.conv2d_f16x2(x1, w3, b, d3, x2=x2, out=y3, gn_groups=32, gn_parts=parts)

        def f1():
            K.conv2d_f16x2(x1, w1, b, d1, x2=x2, out=y1, measure_out=True)

        def both():
            f1(); f3()

        ev_a, ev_b = torch.cuda.Event(), torch.cuda.Event()

        def forked():   # the 1x1 on the second stream between two events, the sibling on the first
            ev_a.record(s1)
            s2.wait_event(ev_a)
            with torch.cuda.stream(s2):
                f1()
                ev_b.record(s2)
            f3()
            s1.wait_event(ev_b)

        with torch.cuda.stream(s2):
            f1()
        torch.cuda.synchronize()
        with torch.cuda.stream(s1):
            ta, tb, tc = timed(f3), timed(f1), timed(both)
            td = timed(forked)
        print(f"{name:18s} {str((t1, k1)):>9s} | {ta:7.2f} {tb:7.2f} {tc:7.2f} {td:9.2f} | {100 * (tc - td) / max(tb, 1e-9):5.1f} %", flush=True)
