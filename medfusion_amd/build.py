"""Builds libmedfusion_hip.so in-tree with hipcc for gfx950 (cross-compiles without a GPU)."""
from __future__ import annotations

import os
import subprocess
import sys
from pathlib import Path

HERE = Path(__file__).resolve().parent
CSRC = HERE / "csrc"
LIB = HERE / "libmedfusion_hip.so"
SOURCES = ["api.hip", "conv.hip", "groupnorm.hip", "small_ops.hip", "sched_noise.hip", "attention.hip", "edge_ops.hip"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-fno-gpu-rdc", "-ffp-contract=off", "-mllvm", "-pragma-unroll-threshold=1000000", "-Wall", "-Wno-unused-function"]


def hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if cand and (os.path.isabs(cand) and os.path.exists(cand) or not os.path.isabs(cand)):
            return cand
    raise RuntimeError("hipcc not found")


def needs_build() -> bool:
    if not LIB.exists():
        return True
    t = LIB.stat().st_mtime
    deps = [CSRC / s for s in SOURCES] + [CSRC / "common.h", CSRC / "gn_partial.h", CSRC / "conv_igemm.h", HERE.parent / "include" / "medfusion_hip.h"]
    return any(d.stat().st_mtime > t for d in deps)


def build(force: bool = False, verbose: bool = True) -> Path:
    if not force and not needs_build():
        return LIB
    cmd = [hipcc(), *FLAGS, *[str(CSRC / s) for s in SOURCES], "-o", str(LIB)]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.run(cmd, check=True)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
    print(LIB)
