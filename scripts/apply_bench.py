#!/usr/bin/env python3
"""Time the GroupNorm-apply pass (mf_gn_apply_from_partials_f32, fp16-pair output, residual + embedding) on the UNet's three levels and the
VAE's largest tensor: microseconds and GB/s of the bytes it has to move (x + residual read, out + pairs written = 16 B / element)."""
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch

from medfusion_amd import kernels as K

dev = torch.device("cuda:0")
for (n, h, w, c, g) in [(16, 32, 32, 256, 32), (16, 16, 16, 512, 32), (16, 8, 8, 1024, 32), (16, 16, 16, 256, 32), (16, 8, 8, 512, 32), (16, 256, 256, 64, 8)]:
    x = torch.randn((n, h, w, c), device=dev)
    res = torch.randn((n, h, w, c), device=dev)
    res._mf_bound = K.maxabs_rows(res)
    emb = torch.randn((n, c), device=dev)
    emb._mf_bound = K.maxabs_rows(emb)
    gamma, beta = torch.randn(c, device=dev), torch.randn(c, device=dev)
    partial, parts = K.gn_stats_partial(x, g)
    rec = K.GnPartials(partial, parts, 1e-5)
    out = torch.empty_like(x)
    for split in (True, False):
        for _ in range(5):
            K.gn_apply(x, rec, gamma, beta, g, 1, res, emb, emb.stride(0), out=out, split=split, bconst=30.0)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        reps = 50
        e0.record()
        for _ in range(reps):
            K.gn_apply(x, rec, gamma, beta, g, 1, res, emb, emb.stride(0), out=out, split=split, bconst=30.0)
        e1.record()
        e1.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / reps
        by = x.numel() * (16 if split else 12)
        print(f"[{n},{h},{w},{c}] parts {parts:3d} split={split}: {us:7.2f} us  {by / 1e6:7.1f} MB  {by / us / 1e3:7.1f} GB/s", flush=True)
