#!/usr/bin/env python3
"""Experiment: does overlapping two half-batches on two HIP streams hide per-launch fixed costs?"""
import sys, time, threading
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch
import medfusion_amd as M
from medfusion_amd import published as P

dev = torch.device("cuda:0")
pipe = P.build_published_pipeline(dev, None)
STEPS = 40

def run(n, seed, stream, out, idx):
    with torch.cuda.stream(stream):
        out[idx] = pipe.sample(n, (8, 32, 32), steps=STEPS, use_ddim=True, noise=M.PhiloxDeviceNoise(seed))

def timed(fn, reps=2):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps

res = [None, None]
s0 = torch.cuda.current_stream()
t1 = timed(lambda: run(16, 1, s0, res, 0))
print(f"1 stream  B=16: {t1*1e3:.1f} ms  -> {16/t1:.2f} img/s (at {STEPS} steps)")
sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
def two():
    ta = threading.Thread(target=run, args=(8, 1, sa, res, 0)); tb = threading.Thread(target=run, args=(8, 2, sb, res, 1))
    ta.start(); tb.start(); ta.join(); tb.join()
t2 = timed(two)
print(f"2 streams B=8+8 (threads): {t2*1e3:.1f} ms -> {16/t2:.2f} img/s")
def four():
    ss = [torch.cuda.Stream() for _ in range(4)]
    r = [None] * 4
    th = [threading.Thread(target=run, args=(4, i, ss[i], r, i)) for i in range(4)]
    [t.start() for t in th]; [t.join() for t in th]
t4 = timed(four)
print(f"4 streams B=4x4 (threads): {t4*1e3:.1f} ms -> {16/t4:.2f} img/s")
t8 = timed(lambda: run(8, 1, s0, res, 0))
print(f"1 stream  B=8: {t8*1e3:.1f} ms  -> {8/t8:.2f} img/s")
t32 = timed(lambda: run(32, 1, s0, res, 0))
print(f"1 stream  B=32: {t32*1e3:.1f} ms  -> {32/t32:.2f} img/s")
