#!/usr/bin/env python3
"""Attribute the sporadic wrong results of round 2's packed-fp32 build of conv_f16x2_kernel to ONE spot of its epilogue.

`build` (runs here, hipcc cross-compiles): diagnostic twins of the library under medfusion_amd/csrc/build/variants/ --
    nopk   the product build (packed fp32 off for conv_f16x2.hip)
    pk     packed fp32 ON, nothing else changed (round 2: sporadic wrong results)
    pk_hzN packed fp32 ON + ONE hook of conv_f16x2.h (-DMFC2_HZ=N):
           1 = 32 wait states between the last matrix instruction and the first VALU read of an accumulator (MFMA -> VALU)
           2 = the data registers of every staging ds_write_b128 stay untouched for 8 states (store-data write-after-read)
           4 = every staging ds_read_b128 fully waited (lgkmcnt(0)) before its first packed consumer
           8 = every hand-off load of the split-K tree waited (vmcnt(0)) before the first add
          16 = pads around the hand-off stores
          32 = the adds of the split-K tree un-packed (v_add_f32 through inline asm)
          64 = the staged value (main + cross / 2048) * scale un-packed
         128 = a bare s_nop 0 behind every staging store (no register operand)
`run` (on the GPU box): scripts/pk_probe.bin, then scripts/conv_stress.py on every twin (one process each); everything goes to stdout.
"""
import os
import subprocess
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
HZ = (1, 2, 4, 8, 16, 32, 64, 128)
PADS = (0, 1, 3)


def build():
    from medfusion_amd import build as B
    print(B.build_variant("nopk"))
    print(B.build_variant("pk", packed_fp32=True))
    for n in HZ:
        print(B.build_variant(f"pk_hz{n}", conv_flags=[f"-DMFC2_HZ={n}"], packed_fp32=True))
    for pad in PADS:   # how long the data registers of the staging ds_write_b128 must stay untouched
        print(B.build_variant(f"pk_hz2_pad{pad}", conv_flags=["-DMFC2_HZ=2", f"-DMFC2_HZ_PAD={pad}"], packed_fp32=True))
    subprocess.check_call([B.hipcc(), "--offload-arch=gfx950", "-O2", str(ROOT / "scripts/pk_probe.hip"), "-o", str(ROOT / "scripts/pk_probe.bin")])


def run(reps):
    vdir = ROOT / "medfusion_amd/csrc/build/variants"
    if not os.environ.get("HUNT_TILES"):
        subprocess.call([str(ROOT / "scripts/pk_probe.bin"), "20000"])
    names = ["nopk", "pk"] + [f"pk_hz{n}" for n in HZ] + [f"pk_hz2_pad{p}" for p in PADS]
    if len(sys.argv) > 3:
        names = sys.argv[3].split(",")
    for name in names:
        env = dict(os.environ, MEDFUSION_LIB=str(vdir / f"libmedfusion_hip_{name}.so"))
        print(f"==== {name}", flush=True)
        subprocess.call([sys.executable, str(ROOT / "scripts/conv_stress.py"), "--reps", str(reps)] + (["--tiles", os.environ["HUNT_TILES"]] if os.environ.get("HUNT_TILES") else []), env=env)
    if os.environ.get("HUNT_TILES"):
        return
    print("==== nopk, MF_CONV_TREE=2 (release / acquire fences around the pair counter)", flush=True)
    subprocess.call([sys.executable, str(ROOT / "scripts/conv_stress.py"), "--reps", str(reps)],
                    env=dict(os.environ, MEDFUSION_LIB=str(vdir / "libmedfusion_hip_nopk.so"), MF_CONV_TREE="2"))


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "build":
        build()
    else:
        run(int(sys.argv[2]) if len(sys.argv) > 2 else 1500)
