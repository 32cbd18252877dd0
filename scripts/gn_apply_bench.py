#!/usr/bin/env python3
"""Timing of the GroupNorm apply pass fed by partial records (mf_gn_apply_from_partials_pairs_f32) on the activation sizes of the published
UNet at B=16, by what the pass has to do: activation or not, embedding, residual (fp32 / fp16 pairs), fp32 output or pairs only.
Run on the GPU box; MEDFUSION_LIB selects a variant build (medfusion_amd.build.build_variant)."""
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch

from medfusion_amd import kernels as K
from _devtime import device_us


def main():
    dev = torch.device("cuda:0")
    g = torch.Generator(device="cpu").manual_seed(0)
    reps = 200
    print(f"{'N,H,W,C':18s} {'case':45s} {'us':>7s} {'GB/s (algorithmic)':>20s}")
    for n, h, c in ((16, 32, 256), (16, 16, 512), (16, 8, 1024), (32, 32, 256)):
        G = 32
        x = torch.randn((n, h, h, c), generator=g).to(dev)
        res = torch.randn((n, h, h, c), generator=g).to(dev)
        emb = torch.randn((n, c), generator=g).to(dev)
        gamma, beta = torch.randn((c,), generator=g).to(dev), torch.randn((c,), generator=g).to(dev)
        partial, parts = K.gn_stats_partial(x, G)
        rec = K.GnPartials(partial, parts, 1e-5)
        bc = float(gamma.abs().max()) * (h * h * c // G) ** 0.5 + float(beta.abs().max())
        for name, kw, nbytes in (
            ("act, emb, fp32+pairs out", dict(act=1, emb=True, out_fp32=True), 12),
            ("act, emb, pairs out", dict(act=1, emb=True, out_fp32=False), 8),
            ("no act, emb, pairs out", dict(act=0, emb=True, out_fp32=False), 8),
            ("act, residual fp32, fp32+pairs", dict(act=1, res=True, out_fp32=True), 16),
            ("act, residual fp32, pairs out", dict(act=1, res=True, out_fp32=False), 12),
            ("no act, residual fp32, pairs", dict(act=0, res=True, out_fp32=False), 12),
        ):
            OUT = torch.empty_like(x)

            def run():
                return K.gn_apply(x, rec, gamma, beta, G, kw["act"], res if kw.get("res") else None, emb if kw.get("emb") else None,
                                  emb.stride(0) if kw.get("emb") else 0, out=OUT, split=True, bconst=bc, out_fp32=kw["out_fp32"])
            us, nl = device_us(run, reps)
            name = f"{name} [{nl} launch]"
            print(f"{n},{h},{h},{c:<10d} {name:45s} {us:7.2f} {x.numel() * nbytes / us / 1e3:20.0f}", flush=True)


if __name__ == "__main__":
    main()
