#!/usr/bin/env python3
"""cProfile of the host side of one sample() at B=1, latent 8x8 (the device is never the bottleneck there): where the 2.4 ms of launch work
per iteration go."""
import cProfile
import pstats
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch

import medfusion_amd as M
from medfusion_amd import published as P

dev = torch.device("cuda:0")
pipe = P.build_published_pipeline(dev, num_classes=None)
pipe.sample(1, (8, 8, 8), steps=20, use_ddim=True, noise=M.PhiloxDeviceNoise(0), use_graph=False)
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
pipe.sample(1, (8, 8, 8), steps=150, use_ddim=True, noise=M.PhiloxDeviceNoise(1), use_graph=False)
torch.cuda.synchronize()
pr.disable()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(28)
st.sort_stats("cumulative").print_stats(30)
