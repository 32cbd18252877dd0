#!/usr/bin/env python3
"""Two identical UNet evaluations at the cfg2 shapes: the first module whose output differs between them (debug aid)."""
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch

import medfusion_amd as M
from medfusion_amd import blocks as BLK
from medfusion_amd import published as P

dev = torch.device("cuda:0")
pipe = P.build_published_pipeline(dev, num_classes=None)
est = pipe.noise_estimator
x = torch.randn((16, 8, 32, 32), device=dev, generator=torch.Generator(device=dev).manual_seed(5))
t = torch.full((16,), 500.0, device=dev)
names = {m: n for n, m in est.named_modules()}
runs = []
for r in range(3):
    rec = []
    hooks = []
    for m in est.modules():
        if isinstance(m, (BLK.Conv, BLK.BasicBlock, BLK.BasicResBlock)):
            def hook(mod, inp, out, rec=rec):
                o = out[0] if isinstance(out, tuple) else out
                rec.append((names[mod], type(mod).__name__, o.detach().clone() if torch.is_tensor(o) else None))
            hooks.append(m.register_forward_hook(hook))
    y, _ = est(x, t, None)
    torch.cuda.synchronize()
    for h in hooks:
        h.remove()
    runs.append(rec)
for r in (1, 2):
    bad = [(a[0], a[1], float((a[2] - b[2]).abs().max()), float(a[2].abs().max())) for a, b in zip(runs[0], runs[r]) if a[2] is not None and not torch.equal(a[2], b[2])]
    print(f"run 0 vs run {r}: {len(bad)} of {len(runs[0])} module outputs differ; first: {bad[:4]}")
