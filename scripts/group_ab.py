#!/usr/bin/env python3
"""Same-process A/B of conv_res inside the launch of its ResBlock's 3x3 (mf_conv2d_f16x2_group; blocks.GROUPED_CONV_RES) on the benchmarked
workload: cfg2 steps with the form off / on, interleaved so that clock and box drift hit both alike; first a bit-equality check of the images.
Writes its verdict (mean ms off / on) as JSON to the path given as argv[1]."""
import json, sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch
import medfusion_amd as M
from medfusion_amd import blocks as BLK, published as P

dev = torch.device("cuda:0")
pipe = P.build_published_pipeline(dev, None)
ROUNDS, STEPS = int(sys.argv[2]) if len(sys.argv) > 2 else 3, 2


def run(steps, seed):
    out = None
    for k in range(steps):
        out = pipe.sample(16, (8, 32, 32), steps=150, use_ddim=True, noise=M.PhiloxDeviceNoise(seed + k))
    torch.cuda.synchronize()
    return out


same, la, lb = True, 0, 0
for bsz, loop in ((16, "cmdlist"), (16, "eager"), (8, "cmdlist")):     # (the pairs that share a launch depend on the batch: 11 at B = 16, 8 at B = 8)
    BLK.GROUPED_CONV_RES = False
    a = pipe.sample(bsz, (8, 32, 32), steps=6, use_ddim=True, noise=M.PhiloxDeviceNoise(7), loop=loop)
    l0 = pipe.last_cmdlist_launches
    BLK.GROUPED_CONV_RES = True
    b = pipe.sample(bsz, (8, 32, 32), steps=6, use_ddim=True, noise=M.PhiloxDeviceNoise(7), loop=loop)
    l1 = pipe.last_cmdlist_launches
    eq = bool(torch.equal(a, b))
    same = same and eq
    if (bsz, loop) == (16, "cmdlist"):
        la, lb = l0, l1
    print(f"B = {bsz}, loop {loop}: images bit-identical: {eq}; launches per recorded iteration {l0} -> {l1}", flush=True)
if ROUNDS == 0:
    sys.exit(0 if same else 1)
res = {False: [], True: []}
for r in range(ROUNDS):
    for on in (False, True):
        BLK.GROUPED_CONV_RES = on
        run(1, 100)
        t0 = time.perf_counter()
        run(STEPS, 200 + 10 * r)
        res[on].append((time.perf_counter() - t0) / STEPS * 1e3)
off, on = sum(res[False]) / ROUNDS, sum(res[True]) / ROUNDS
print(f"cfg2 (B = 16, 150 DDIM iterations + decode), ms per step over {ROUNDS} interleaved rounds of {STEPS} steps:")
for k, name in ((False, "two launches"), (True, "one launch")):
    v = res[k]
    print(f"  {name:14s} " + " ".join(f"{x:7.2f}" for x in v) + f"   mean {sum(v) / len(v):7.2f} ms = {16e3 / (sum(v) / len(v)):6.2f} images/s")
print(f"  one launch vs two: {100 * (off / on - 1):+.2f} %  (launches per iteration {la} -> {lb})")
if len(sys.argv) > 1:
    Path(sys.argv[1]).write_text(json.dumps({"bit_identical": same, "ms_two_launches": off, "ms_one_launch": on, "gain_pct": 100 * (off / on - 1), "launches": [la, lb]}))
