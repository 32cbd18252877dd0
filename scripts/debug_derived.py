#!/usr/bin/env python3
"""debug: published UNet forward at B=2 with derived-bound pair outputs on / off, layer by layer"""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch
import medfusion_amd as M
from medfusion_amd import blocks as BLK, kernels as K, published as P

dev = torch.device("cuda:0")
pipe = P.build_published_pipeline(dev, 2)
est = pipe.noise_estimator
torch.manual_seed(0)
x = torch.randn(2, 8, 32, 32, device=dev)
t = torch.tensor([731, 731], device=dev)
c = torch.tensor([1, 0], device=dev)


def decode(t_):
    s = t_._mf_split
    raw = s.view(torch.int16).view(*t_.shape[:-1], t_.shape[-1] // 8, 2, 8)
    hi = raw[..., 0, :].contiguous().view(torch.float16).double().reshape(t_.shape)
    lo = raw[..., 1, :].contiguous().view(torch.float16).double().reshape(t_.shape)
    sc = torch.exp2(torch.floor(torch.log2(t_._mf_bound.double())) - 14).view(-1, 1, 1, 1)
    return (hi + lo / 2048.0) * sc


def run(flag):
    BLK.DERIVED_OUT_BOUNDS = flag
    rec = []
    hooks = []
    for name, m in est.named_modules():
        if isinstance(m, BLK.Conv):
            def hk(mod, inp, out, name=name):
                o = out[0] if isinstance(out, tuple) else out
                if not torch.is_tensor(o):
                    return
                info = {"name": name, "y": o.detach().clone(), "pairs": None}
                if getattr(o, "_mf_split", None) is not None and getattr(o, "_mf_bound", None) is not None and o.dim() == 4:
                    info["pairs"] = decode(o)
                    info["bound"] = o._mf_bound.clone()
                rec.append(info)
            hooks.append(m.register_forward_hook(hk))
    y, _ = est(x, t, c)
    torch.cuda.synchronize()
    for h in hooks:
        h.remove()
    return y, rec


y1, r1 = run(True)
y0, r0 = run(False)
print("final rel diff", float((y1 - y0).abs().max() / y0.abs().max()))
for a, b in zip(r1, r0):
    d = float((a["y"] - b["y"]).abs().max() / b["y"].abs().max().clamp_min(1e-30))
    extra = ""
    if a["pairs"] is not None:
        e = float((a["pairs"] - a["y"].double()).abs().max() / a["y"].abs().max())
        extra = f" | pairs-vs-fp32 {e:.2e} bound {a['bound'].tolist()} max|y| {a['y'].abs().amax(dim=(1,2,3)).tolist()}"
    print(f"{a['name']:45s} shape {tuple(a['y'].shape)} diff(on,off) {d:.2e}{extra}")
    if d > 1e-3:
        break
