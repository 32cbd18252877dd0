#!/usr/bin/env python3
"""Tune the (tile, split-K) of the UNet's convolution shapes INSIDE the real sampling loop (round 4).  scripts/conv_sweep.py times a shape in
isolation (back-to-back launches of one convolution); its ranking is right to a few per cent, but the loop has other cache and clock
states -- tile 34 against tile 52 at 32 x 32 was 7 % slower in the sweep and 0.2 % FASTER in the loop.  This script takes the sweep's top
candidates per shape (its output file), installs each as a run-time plan override (mf_conv2d_plan_override), and times whole cfg2 samples.
A candidate is kept when it beats the current plan in BOTH of two interleaved rounds by more than the noise.
usage: plan_tune.py <sweep output file> [--top 3] [--reps 2]"""
import argparse, ctypes as C, re, sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
sys.path.insert(0, str(Path(__file__).resolve().parent))
import torch
import medfusion_amd as M
from medfusion_amd import kernels as K, lib as L, published as P
from conv_sweep import unet_shapes

ap = argparse.ArgumentParser()
ap.add_argument("sweep")
ap.add_argument("--top", type=int, default=3)
ap.add_argument("--reps", type=int, default=2)
ap.add_argument("--batch", type=int, default=16)
ap.add_argument("--latent", type=int, default=32, help="latent size (64: the 512-px workload)")
a = ap.parse_args()
dev = torch.device("cuda:0")
pipe = P.build_published_pipeline(dev, None)
lib = L.load()
f_ = a.latent // 32
shapes = {s[0]: (s[0], s[1], s[2] * f_, s[3] * f_, *s[4:]) for s in unet_shapes(a.batch)}
cands = {}
for ln in Path(a.sweep).read_text().splitlines():
    parts = ln.split("|")
    if len(parts) < 4:
        continue
    name = parts[0][:24].strip()
    if name not in shapes:
        continue
    allres = re.findall(r"(\d+)/(\d+):([0-9.]+)", parts[3])
    cands[name] = [(int(t), int(s)) for t, s, _ in allres[: a.top]]


def desc_of(name):
    _, n, h, w, c1, c2, co, k, st, ups, _ = shapes[name]
    return K.make_conv_desc(n, h, w, c1, c2, co, k, st, 1 if k == 3 else 0, 2 if ups else 0, precision=5)


def clear_caches():
    for m in pipe.modules():
        if hasattr(m, "_descs"):
            m._descs.clear()
            m._pairs_out.clear()


def run(nsteps, seed):
    for k in range(nsteps):
        pipe.sample(a.batch, (8, a.latent, a.latent), steps=150, use_ddim=True, noise=M.PhiloxDeviceNoise(seed + k), decode=False)
    torch.cuda.synchronize()


def timed():
    clear_caches()
    pipe.sample(a.batch, (8, a.latent, a.latent), steps=12, use_ddim=True, noise=M.PhiloxDeviceNoise(1), decode=False)   # plans, workspaces, command list
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    run(a.reps, 100)
    return (time.perf_counter() - t0) / a.reps * 1e3


print(f"in-loop plan tuning, cfg2 denoise loop (B = {a.batch}, 150 iterations, no decode), ms per sample() over {a.reps} runs; candidates: top {a.top} of {a.sweep}")
keep = []
seen_keys = set()
for name, cl in cands.items():
    d = desc_of(name)
    key = (d.N, d.Hin, d.Win, d.C1 + d.C2, d.Cout, d.KH, d.stride, d.upsample)
    if key in seen_keys:      # (c0 / c1 of different blocks share a shape: one plan)
        continue
    seen_keys.add(key)
    if d.KH == 3 and d.stride == 1 and d.upsample == 0 and K.wino_preferred(d):   # (round 5: this shape runs on its Winograd form -- its direct plan is not in the loop)
        continue
    lib.mf_conv2d_plan_override(C.byref(d), 0, 0)
    cur = K.conv_plan(d)
    res = {}
    for rnd in range(2):
        for t, s in [cur] + [c for c in cl if c != cur]:
            lib.mf_conv2d_plan_override(C.byref(d), t, s)
            res.setdefault((t, s), []).append(timed())
    lib.mf_conv2d_plan_override(C.byref(d), 0, 0)
    base = res[cur]
    line = f"{name:22s} current {cur}: {base[0]:7.2f} {base[1]:7.2f} |"
    best = None
    for c, v in res.items():
        if c == cur:
            continue
        line += f" {c}: {v[0]:7.2f} {v[1]:7.2f}"
        if v[0] < base[0] * 0.9985 and v[1] < base[1] * 0.9985 and (best is None or sum(v) < sum(res[best])):
            best = c
    if best:
        line += f"  -> {best} ({100 * (sum(base) / sum(res[best]) - 1):+.2f} %)"
        keep.append((key, best))
        lib.mf_conv2d_plan_override(C.byref(d), best[0], best[1])     # keep it installed: later shapes are tuned on top of it
    print(line, flush=True)
print("table entries to change:")
for key, (t, s) in keep:
    print("    {%s, %d, %d}," % (", ".join(str(v) for v in key), t, s))
