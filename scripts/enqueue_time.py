#!/usr/bin/env python3
"""How far ahead of the GPU is the host?  Times one cfg2 sample(): the moment the Python call returns (everything enqueued) and the
moment the device is idle.  If the two are close the loop is launch-bound."""
import sys
import time
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch

import medfusion_amd as M
from medfusion_amd import published as P

dev = torch.device("cuda:0")
pipe = P.build_published_pipeline(dev, num_classes=None)
pipe.sample(16, (8, 32, 32), steps=150, use_ddim=True, noise=M.PhiloxDeviceNoise(0))   # warm-up
rep = 0
for b, lat in ((16, 32), (8, 32), (4, 32), (1, 8)):
    for loop in ("eager", "graph", "cmdlist", "cmdlist"):
        rep += 1
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        img = pipe.sample(b, (8, lat, lat), steps=150, use_ddim=True, noise=M.PhiloxDeviceNoise(rep), loop=loop)
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        print(f"B={b} latent {lat} loop={loop:8s}: enqueued after {1e3 * (t1 - t0):6.1f} ms ({1e3 * (t1 - t0) / 150:.2f} ms / iteration), device idle after "
              f"{1e3 * (t2 - t0):6.1f} ms", flush=True)
    # the host side of ONE replayed iteration with nothing queued in front of it (a launch call blocks once the hardware queue is full,
    # so "enqueued after" above is the DEVICE's pace whenever the device is the slower side)
    pipe.time_cmdlist = True
    pipe.sample(b, (8, lat, lat), steps=20, use_ddim=True, noise=M.PhiloxDeviceNoise(99), loop="cmdlist")
    pipe.time_cmdlist = False
    print(f"B={b} latent {lat}: command list of {pipe.last_cmdlist_launches} launches, host time of one replayed iteration {pipe.last_cmdlist_host_ms:.3f} ms "
          f"({1e3 * pipe.last_cmdlist_host_ms / max(1, pipe.last_cmdlist_launches):.2f} us per launch)", flush=True)
