#!/usr/bin/env python3
"""Throughput of mf_attention_f32 at the published-width UNet levels (use_attention='spatial': 8 heads, d = C/8)."""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch
from medfusion_amd import kernels as K

dev = torch.device("cuda:0")
for (b, h, n, d) in ((16, 8, 1024, 32), (16, 8, 256, 64), (16, 8, 64, 128), (8, 8, 4096, 32)):
    c = h * d
    q, k, v = (torch.randn((b, n, c), device=dev) for _ in range(3))
    K.attention(q, k, v, h, d ** -0.25)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        K.attention(q, k, v, h, d ** -0.25)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    gf = 4.0 * b * h * n * n * d / 1e9
    print(f"B={b} H={h} N={n} d={d} (C={c}): {ms:.3f} ms  {gf / ms:.1f} TF = {gf / ms / 157.3:.3f} of the 157.3 TF fp32-MFMA peak  ({gf:.1f} GF: QK^T + PV, fp32 operands)")
