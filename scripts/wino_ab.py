#!/usr/bin/env python3
"""Same-process A/B of the Winograd form on the benchmarked workload: cfg2 steps with blocks.WINOGRAD = 0 / 1 (shapes: csrc/wino_plan_table.inc
plus MEDFUSION_WINOGRAD_TABLE), interleaved so that clock and box drift hit both alike; first the difference of the images (NOT bit-identical:
another summation) and the 150-iteration trajectory error of both forms against each other.  argv[1]: JSON verdict, argv[2]: rounds, argv[3]: batch."""
import json, sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch
import medfusion_amd as M
from medfusion_amd import blocks as BLK, published as P

dev = torch.device("cuda:0")
pipe = P.build_published_pipeline(dev, None)
ROUNDS, STEPS = int(sys.argv[2]) if len(sys.argv) > 2 else 3, 2
B = int(sys.argv[3]) if len(sys.argv) > 3 else 16


def run(steps, seed):
    out = None
    for k in range(steps):
        out = pipe.sample(B, (8, 32, 32), steps=150, use_ddim=True, noise=M.PhiloxDeviceNoise(seed + k))
    torch.cuda.synchronize()
    return out


imgs, launches = {}, {}
for mode in (0, 1):
    BLK.WINOGRAD = mode
    imgs[mode] = pipe.sample(B, (8, 32, 32), steps=150, use_ddim=True, noise=M.PhiloxDeviceNoise(7)).clone()
    launches[mode] = pipe.last_cmdlist_launches
diff = float((imgs[1] - imgs[0]).abs().max() / imgs[0].abs().max())
print(f"B = {B}: 150-iteration images, Winograd vs direct: max-norm rel diff {diff:.2e}; launches per recorded iteration {launches[0]} -> {launches[1]}", flush=True)
res = {0: [], 1: []}
for r in range(ROUNDS):
    for mode in (0, 1):
        BLK.WINOGRAD = mode
        run(1, 100)
        t0 = time.perf_counter()
        run(STEPS, 200 + 10 * r)
        res[mode].append((time.perf_counter() - t0) / STEPS * 1e3)
off, on = sum(res[0]) / ROUNDS, sum(res[1]) / ROUNDS
print(f"cfg2-style (B = {B}, 150 DDIM iterations + decode), ms per step over {ROUNDS} interleaved rounds of {STEPS} steps:")
for k, name in ((0, "direct"), (1, "winograd")):
    v = res[k]
    print(f"  {name:10s} " + " ".join(f"{x:7.2f}" for x in v) + f"   mean {sum(v) / len(v):7.2f} ms = {B * 1e3 / (sum(v) / len(v)):6.2f} images/s")
print(f"  winograd vs direct: {100 * (off / on - 1):+.2f} %")
if len(sys.argv) > 1:
    Path(sys.argv[1]).write_text(json.dumps({"rel_diff_images": diff, "ms_direct": off, "ms_winograd": on, "gain_pct": 100 * (off / on - 1), "launches": [launches[0], launches[1]]}))
