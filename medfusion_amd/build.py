"""Builds libmedfusion_hip.so in-tree with hipcc for gfx950 (cross-compiles without a GPU).

One object per source (compiled in parallel, rebuilt only when the source or a header changed), then one link.
"""
from __future__ import annotations

import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor
from pathlib import Path

HERE = Path(__file__).resolve().parent
CSRC = HERE / "csrc"
OBJ = CSRC / "build"
LIB = HERE / "libmedfusion_hip.so"
SOURCES = ["api.hip", "conv.hip", "conv_f16x2.hip", "groupnorm.hip", "small_ops.hip", "sched_noise.hip", "attention.hip", "edge_ops.hip"]
HEADERS = ["common.h", "gn_partial.h", "conv_igemm.h", "conv_f16x2.h", "conv_f16x2_body.inc", "conv_f16x2_halo.h", "conv_f16x2_halo_body.inc", "conv_f16x2_group.h", "conv_f16x2_epilogue.inc", "conv_plan.h", "split_f16.h", "conv_plan_table.inc", "winograd.h", "conv_f16x2_wino.inc", "wino_plan_table.inc"]
# -packed-fp32-ops for conv_f16x2.hip.  gfx950 erratum, root-caused in round 3 (profiles/r03_pk_repro.txt, scripts/pk_repro_min.hip): a packed
# fp32 instruction whose LOW result takes the HIGH half of src1 ("v_pk_mul_f32 vD, vA, vB op_sel:[0,1]") reads that operand as 0.0 in lanes
# 48..63 now and then, while the other wave of the SIMD issues MFMAs and LDS reads return -- exactly what a co-resident conv workgroup does.
# hipcc 7.2 forms that selection by itself when it SLP-packs "(main + cross / 2048) * scale" (72 instructions in the packed build), and a
# K-slice partial became 0 in 16 lanes of a register (round 2: "a partner's split-K tile was not added", "the bare bias").  The convolution
# unit is therefore built without packed fp32 at all (32.0 vs 32.1 images/s); the other units keep it (the fp32 edge convolutions are 40 %
# faster with it) and are scanned for the failing operand selection by lint_isa() below (CPU test).  (A per-kernel
# `__attribute__((target("no-packed-fp32-ops")))` cost 10 % of the step -- helpers without the attribute are no longer inlined.)
CFLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fno-gpu-rdc", "-ffp-contract=off", "-mllvm", "-pragma-unroll-threshold=1000000", "-Wall",
          "-Wno-unused-function"]
EXTRA_CFLAGS = {"conv_f16x2.hip": ["-Xclang", "-target-feature", "-Xclang", "-packed-fp32-ops"]}
LFLAGS = ["--offload-arch=gfx950", "-shared", "-fPIC", "-fno-gpu-rdc"]


def hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if cand and (os.path.isabs(cand) and os.path.exists(cand) or not os.path.isabs(cand)):
            return cand
    raise RuntimeError("hipcc not found")


def _deps():
    return [CSRC / h for h in HEADERS] + [HERE.parent / "include" / "medfusion_hip.h"]


def _stale(target: Path, deps) -> bool:
    if not target.exists():
        return True
    t = target.stat().st_mtime
    return any(d.stat().st_mtime > t for d in deps)


def needs_build() -> bool:
    return _stale(LIB, [CSRC / s for s in SOURCES] + _deps())


def _run(cmd, verbose):
    if verbose:
        print(" ".join(cmd), flush=True)
    r = subprocess.run(cmd, stderr=subprocess.PIPE, text=True)
    # (the HOST pass of a -c compile does not know the device feature named in CFLAGS and says so three times per file)
    err = "\n".join(line for line in r.stderr.splitlines()
                    if "is not a recognized feature for this target" not in line and "argument unused during compilation: '--hip-link'" not in line)
    if err.strip():
        print(err, file=sys.stderr, flush=True)
    if r.returncode:
        raise subprocess.CalledProcessError(r.returncode, cmd)


def _link(objs, lib: Path, verbose: bool) -> None:
    """link to a temporary name, load it (every symbol must resolve: a kernel stub the host pass dropped shows up here, not on the GPU
    box), then rename onto `lib` -- a process that dlopens `lib` meanwhile sees the old file or the new one, never a half-written one"""
    import ctypes
    tmp = lib.with_name(f".{lib.name}.{os.getpid()}.tmp")
    try:
        _run([hipcc(), *LFLAGS, *[str(o) for o in objs], "-o", str(tmp)], verbose)
        ctypes.CDLL(str(tmp))
        os.replace(tmp, lib)
    finally:
        if tmp.exists():
            tmp.unlink()


def build(force: bool = False, verbose: bool = True) -> Path:
    if not force and not needs_build():
        return LIB
    OBJ.mkdir(exist_ok=True)
    cc = hipcc()
    jobs = []
    for src in SOURCES:
        obj = OBJ / (Path(src).stem + ".o")
        if force or _stale(obj, [CSRC / src] + _deps()):
            jobs.append([cc, *CFLAGS, *EXTRA_CFLAGS.get(src, []), "-c", str(CSRC / src), "-o", str(obj)])
    with ThreadPoolExecutor(max_workers=min(8, max(1, len(jobs)))) as ex:
        list(ex.map(lambda c: _run(c, verbose), jobs))
    # the packed-fp32 erratum guard: the ISA lint is part of the build, not only of a CPU test -- a compiler or source change that
    # brings `op_sel:[x,1]` back into a shipped unit fails HERE (the -S outputs are cached next to the objects)
    bad = lint_isa(verbose=False)
    if bad:
        raise RuntimeError("medfusion_amd.build: packed fp32 instructions that take the HIGH half of src1 for their low result (gfx950 erratum, "
                           f"profiles/r03_pk_repro.txt) in: {sorted(set(b[0] for b in bad))} -- e.g. {bad[0][1]!r}")
    _link([OBJ / (Path(s).stem + ".o") for s in SOURCES], LIB, verbose)
    return LIB


def conv_source_stamp() -> str:
    """16 hex digits that change whenever the fp16-pair convolution's sources, its tile table or its build flags change: the PMC traffic
    file (profiles/pmc_bench_traffic.json) carries the stamp of the tree it was measured on, bench.py compares (roofline.traffic_stale)."""
    import hashlib
    h = hashlib.sha256()
    for name in ("conv_f16x2.h", "conv_f16x2_body.inc", "conv_f16x2.hip", "conv_f16x2_epilogue.inc", "conv_f16x2_halo.h", "conv_f16x2_halo_body.inc",
                 "conv_f16x2_group.h", "conv_plan_table.inc", "split_f16.h", "winograd.h", "conv_f16x2_wino.inc", "wino_plan_table.inc"):
        h.update(name.encode())
        h.update((CSRC / name).read_bytes())
    h.update(" ".join(CFLAGS + EXTRA_CFLAGS.get("conv_f16x2.hip", [])).encode())
    return h.hexdigest()[:16]


def build_variant(name: str, conv_flags=(), packed_fp32: bool = False, verbose: bool = False, unit_flags=None) -> Path:
    """A diagnostic twin of the library (csrc/build/variants/libmedfusion_hip_<name>.so; MEDFUSION_LIB=<path> makes lib.load() take it):
    conv_f16x2.hip recompiled with `conv_flags` (e.g. -DMFC2_HZ=2) and, with packed_fp32, WITHOUT the -packed-fp32-ops switch-off;
    `unit_flags` = {"groupnorm.hip": ["-DX=1"], ...} recompiles other units with extra flags; every other object is the product
    build's.  scripts/pk_hunt.py uses it to attribute a wrong result to one spot of the epilogue, the A/B timing scripts for experiments."""
    build(verbose=verbose)
    vdir = OBJ / "variants"
    vdir.mkdir(exist_ok=True)
    flags = {k: list(v) for k, v in (unit_flags or {}).items()}
    if conv_flags or packed_fp32 or not flags:
        flags["conv_f16x2.hip"] = list(conv_flags) + flags.get("conv_f16x2.hip", [])
    objs = {}
    for src, fl in flags.items():
        obj = vdir / f"{Path(src).stem}_{name}.o"
        extra = [] if (packed_fp32 and src == "conv_f16x2.hip") else EXTRA_CFLAGS.get(src, [])
        _run([hipcc(), *CFLAGS, *extra, *fl, "-c", str(CSRC / src), "-o", str(obj)], verbose)
        objs[src] = obj
    lib = vdir / f"libmedfusion_hip_{name}.so"
    _link([objs.get(s, OBJ / (Path(s).stem + ".o")) for s in SOURCES], lib, verbose)
    return lib


def asan_runtime() -> Path:
    """the AddressSanitizer runtime of the ROCm clang (to LD_PRELOAD in front of a python that loads the ASAN twin)"""
    clang = Path(hipcc()).resolve().parent.parent / "lib" / "llvm" / "bin" / "clang"
    if not clang.exists():
        clang = Path("/opt/rocm/lib/llvm/bin/clang")
    out = subprocess.run([str(clang), "-print-file-name=libclang_rt.asan-x86_64.so"], capture_output=True, text=True, check=True).stdout.strip()
    return Path(out)


def build_asan(verbose: bool = False) -> Path:
    """The HOST side of the library under AddressSanitizer (sanitizers row): csrc/build/variants/libmedfusion_hip_asan.so -- every
    unit recompiled with -fsanitize=address for the host pass only (planner, descriptor validation, workspace arithmetic, command lists, the timing
    registry: the code a wrong struct layout or a bad size would corrupt), device code unchanged.  Use:
        LD_PRELOAD=$(python -c "from medfusion_amd.build import asan_runtime; print(asan_runtime())") ASAN_OPTIONS=detect_leaks=0 \
        MEDFUSION_LIB=<the path this returns> python -m pytest tests -m "not gpu"
    (tests/test_boundary_cpu.py::test_host_side_under_address_sanitizer runs the planner and boundary tests that way)."""
    vdir = OBJ / "variants"
    vdir.mkdir(parents=True, exist_ok=True)
    lib = vdir / "libmedfusion_hip_asan.so"
    if not _stale(lib, [CSRC / s for s in SOURCES] + _deps() + [Path(__file__)]):
        return lib
    san = ["-Xarch_host", "-fsanitize=address", "-Xarch_host", "-fno-omit-frame-pointer", "-g"]
    jobs, objs = [], []
    for src in SOURCES:
        obj = vdir / f"{Path(src).stem}_asan.o"
        objs.append(obj)
        jobs.append([hipcc(), *CFLAGS, *EXTRA_CFLAGS.get(src, []), *san, "-c", str(CSRC / src), "-o", str(obj)])
    with ThreadPoolExecutor(max_workers=min(8, len(jobs))) as ex:
        list(ex.map(lambda c: _run(c, verbose), jobs))
    tmp = lib.with_name(f".{lib.name}.{os.getpid()}.tmp")
    try:
        _run([hipcc(), *LFLAGS, "-fsanitize=address", "-shared-libsan", *[str(o) for o in objs], "-o", str(tmp)], verbose)
        os.replace(tmp, lib)      # (not loaded here: it needs its runtime preloaded)
    finally:
        if tmp.exists():
            tmp.unlink()
    return lib


# gfx950 erratum (scripts/pk_repro_min.hip, profiles/r03_pk_repro.txt): a packed fp32 VALU instruction whose LOW result takes the HIGH half of
# src1 (op_sel's second bit set: "v_pk_mul_f32 vD, vA, vB op_sel:[0,1]") reads that operand as 0.0 in lanes 48..63 now and then, when the
# other wave of the SIMD issues matrix instructions while LDS reads return.  hipcc 7.2 forms that operand selection by itself when it packs
# scalar fp32 code, and pads nothing.  No translation unit of the library may contain it.
_PK_SRC1_HIGH = __import__("re").compile(r"^\s*v_pk_(?:mul|add|fma)_f32\b.*\bop_sel:\[[01],1")


def lint_isa(verbose: bool = False):
    """Compile every translation unit to device assembly with its own flags (cached under csrc/build/lint) and return the packed fp32
    instructions that select the high half of src1 for their low result, as (source, line text) pairs -- must be empty."""
    lint = OBJ / "lint"
    lint.mkdir(parents=True, exist_ok=True)
    cc = hipcc()

    def one(src):
        out = lint / (Path(src).stem + ".s")
        if _stale(out, [CSRC / src] + _deps() + [Path(__file__)]):
            _run([cc, *CFLAGS, *EXTRA_CFLAGS.get(src, []), "--cuda-device-only", "-S", str(CSRC / src), "-o", str(out)], verbose)
        return [(src, ln.strip()) for ln in out.read_text().splitlines() if _PK_SRC1_HIGH.match(ln)]

    with ThreadPoolExecutor(max_workers=8) as ex:
        return [hit for hits in ex.map(one, SOURCES) for hit in hits]


if __name__ == "__main__":
    build(force="--force" in sys.argv)
    print(LIB)
    if "--asan" in sys.argv:
        print(build_asan(verbose=True), "(runtime to preload:", asan_runtime(), ")")
    if "--lint" in sys.argv:
        bad = lint_isa()
        print("ISA lint:", "clean" if not bad else bad)
