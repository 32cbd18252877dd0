#!/usr/bin/env python3
"""A/B of library / host switches that are read once per process (MF_* / MEDFUSION_* environment variables) on the benchmarked workload, ON ONE BOX:
one child process per setting and round, the settings interleaved so that box and clock drift hit all alike.
    python scripts/env_ab.py [--rounds 3] [--batch 16] [--steps 150] "A=0" "A=1 B=2" ...       (an empty string "" = the defaults)
Each child: build the published pipeline, one warm-up sample(), then 2 timed sample() calls; prints ms per step."""
import argparse
import os
import subprocess
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]


def child(batch, steps):
    sys.path.insert(0, str(ROOT))
    import torch
    import medfusion_amd as M
    from medfusion_amd import published as P
    dev = torch.device("cuda:0")
    pipe = P.build_published_pipeline(dev, None)
    pipe.sample(batch, (8, 32, 32), steps=steps, use_ddim=True, noise=M.PhiloxDeviceNoise(1))
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(2):
        pipe.sample(batch, (8, 32, 32), steps=steps, use_ddim=True, noise=M.PhiloxDeviceNoise(10 + k))
    torch.cuda.synchronize()
    print(f"MS_PER_STEP {(time.perf_counter() - t0) / 2 * 1e3:.3f}", flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rounds", type=int, default=3)
    ap.add_argument("--batch", type=int, default=16)
    ap.add_argument("--steps", type=int, default=150)
    ap.add_argument("--child", action="store_true")
    ap.add_argument("settings", nargs="*")
    a = ap.parse_args()
    if a.child:
        return child(a.batch, a.steps)
    res = {s: [] for s in a.settings}
    for r in range(a.rounds):
        for s in a.settings:
            env = dict(os.environ)
            for kv in s.split():
                k, _, v = kv.partition("=")
                env[k] = v
            out = subprocess.run([sys.executable, __file__, "--child", "--batch", str(a.batch), "--steps", str(a.steps)], env=env, capture_output=True, text=True)
            ms = [float(ln.split()[1]) for ln in out.stdout.splitlines() if ln.startswith("MS_PER_STEP")]
            if not ms:
                print(f"[{s!r}] child failed:\n{out.stderr[-800:]}", flush=True)
                continue
            res[s].append(ms[0])
    base = None
    print(f"B = {a.batch}, {a.steps} DDIM iterations + decode, ms per sample() over {a.rounds} interleaved rounds (one process per setting and round):")
    for s in a.settings:
        v = res[s]
        if not v:
            continue
        m = sum(v) / len(v)
        base = base or m
        print(f"  {s or '(defaults)':44s} " + " ".join(f"{x:8.2f}" for x in v) + f"   mean {m:8.2f} ms = {a.batch * 1e3 / m:6.2f} images/s  ({100 * (base / m - 1):+.2f} % vs the first)")


if __name__ == "__main__":
    main()
