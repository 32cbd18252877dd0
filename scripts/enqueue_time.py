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
for rep, (b, lat, graph) in enumerate([(16, 32, False), (16, 32, False), (16, 32, True), (16, 32, True), (1, 8, False), (1, 8, True), (4, 32, False), (4, 32, True),
                                       (8, 32, False), (8, 32, True)]):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    img = pipe.sample(b, (8, lat, lat), steps=150, use_ddim=True, noise=M.PhiloxDeviceNoise(rep), use_graph=graph)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"rep {rep} B={b} latent {lat} graph={graph}: enqueued after {1e3 * (t1 - t0):.1f} ms, device idle after {1e3 * (t2 - t0):.1f} ms", flush=True)
