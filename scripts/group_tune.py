#!/usr/bin/env python3
"""Tune the grouped launches (conv_res inside the launch of its ResBlock's 3x3, mf_conv2d_f16x2_group) INSIDE the real sampling loop: per
channel-changing ResBlock of the UNet, time whole denoise loops with the pair as two launches, with the guest tile the host logic picks, and with
every other guest tile the pair is instantiated for (blocks.GROUP_GUEST is the override hook).  A choice is reported when it beats the current
one in BOTH of two interleaved rounds by more than the noise.
usage: group_tune.py [--batch 16] [--latent 32] [--reps 2]"""
import argparse, sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch
import medfusion_amd as M
from medfusion_amd import blocks as BLK, kernels as K, published as P

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=16)
ap.add_argument("--latent", type=int, default=32)
ap.add_argument("--reps", type=int, default=2)
a = ap.parse_args()
dev = torch.device("cuda:0")
pipe = P.build_published_pipeline(dev, None)
blocks = [m for m in pipe.modules() if isinstance(m, BLK.BasicResBlock) and not isinstance(m.conv_res, torch.nn.Identity)]


def clear():
    for m in blocks:
        m._group.clear()


def run(n, seed):
    for k in range(n):
        pipe.sample(a.batch, (8, a.latent, a.latent), steps=150, use_ddim=True, noise=M.PhiloxDeviceNoise(seed + k), decode=False)
    torch.cuda.synchronize()


def timed():
    clear()
    pipe.sample(a.batch, (8, a.latent, a.latent), steps=12, use_ddim=True, noise=M.PhiloxDeviceNoise(1), decode=False)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    run(a.reps, 100)
    return (time.perf_counter() - t0) / a.reps * 1e3


timed()
keys = {}
for m in blocks:           # the shapes the loop really ran (filled by the warm-up above)
    for k, ent in m._group.items():
        n, h, w, c1, c2 = k
        key = (n, h, w, c1, c2, m.conv_res.out_ch)
        keys.setdefault(key, None if ent is None else (K.conv_plan(ent[0]), K.conv_plan(ent[3])))
print(f"grouped-launch tuning, denoise loop (B = {a.batch}, latent {a.latent}, 150 iterations, no decode), ms per sample() over {a.reps} runs")
for key, cur in keys.items():
    res = {}
    for rnd in range(2):
        for choice in (None, -1, 36, 37, 53, (36, 1), (36, 2), (36, 4), (37, 1), (37, 2), (53, 1), (53, 2)):
            if choice is None:
                BLK.GROUP_GUEST.pop(key, None)
            else:
                BLK.GROUP_GUEST[key] = choice
            clear()
            if choice not in (None, -1):     # is this pair possible at all?  (ask the host logic on a block of that shape)
                blk = next(m for m in blocks if m.conv_res.out_ch == key[5] and m.conv_res.in_ch == key[3] + key[4])
                x1 = torch.empty((key[0], key[1], key[2], key[3]), device="meta")
                x = x1 if not key[4] else (x1, torch.empty((key[0], key[1], key[2], key[4]), device="meta"))
                ok = blk._grouped(x) is not None
                clear()
                if not ok:
                    continue
            res.setdefault(choice, []).append(timed())
    BLK.GROUP_GUEST.pop(key, None)
    base = res[None]
    line = f"{key}: current {cur}: {base[0]:7.2f} {base[1]:7.2f} |"
    best = None
    for c, v in res.items():
        if c is None:
            continue
        line += f" {'two launches' if c == -1 else 'guest %s' % (c,)}: {v[0]:7.2f} {v[1]:7.2f}"
        if v[0] < base[0] * 0.9985 and v[1] < base[1] * 0.9985 and (best is None or sum(v) < sum(res[best])):
            best = c
    if best is not None:
        line += f"  -> {'two launches' if best == -1 else 'guest %s' % (best,)} ({100 * (sum(base) / sum(res[best]) - 1):+.2f} %)"
        BLK.GROUP_GUEST[key] = best      # keep it: later blocks are tuned on top of it
    print(line, flush=True)
print("overrides that beat the host logic:", dict(BLK.GROUP_GUEST))
