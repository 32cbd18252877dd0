"""The C-ABI boundary without a GPU: the library loads, exports and binds every symbol include/*.h declares,
the product never touches the oracle, and there is no CPU fallback."""
import ast
import re
from pathlib import Path

import pytest
import torch

ROOT = Path(__file__).resolve().parents[1]


def test_library_loads_and_exports_every_declared_symbol():
    from medfusion_amd import lib as L
    h = (ROOT / "include" / "medfusion_hip.h").read_text()
    declared = set(re.findall(r"\b(mf_[a-z0-9_]+)\s*\(", h))
    assert len(declared) >= 25
    lib = L.load()
    for name in sorted(declared):
        assert hasattr(lib, name), f"{name} declared in the header but not exported"
    assert declared == set(L.exported_symbols()), declared ^ set(L.exported_symbols())
    assert lib.mf_version() == 250
    assert lib.mf_prof_family_name(0) == b"conv_igemm"


def test_struct_layouts_match_header():
    import ctypes as C
    from medfusion_amd import lib as L
    assert C.sizeof(L.MfConvDesc) == 16 * 4
    assert C.sizeof(L.MfSchedStep) == 12 * 4
    assert C.sizeof(L.MfSchedArgs) == 8 * 6 + 8 + 8 * 5 + 4 * 4 + 8  # 6 ptr, i64, 5 ptr, 3 i32 + f32, i64
    assert C.sizeof(L.MfGnFuse) == 8 * 13 + 8 + 4 * 4                   # 13 ptr, i64, 2 i32 + 2 f32
    assert C.sizeof(L.MfConvF16x2Call) == 8 * 7 + 8 + 8 * 5 + 8 + 8           # 7 ptr, f32 (+ pad), 5 ptr / size_t, i32 (+ pad), ptr
    assert C.sizeof(L.MfWinoTail) == 8 * 13 + 8 + 4 * 4                 # 13 ptr, i64, 2 i32 + 2 f32


def test_struct_layouts_match_what_a_c_compiler_sees(tmp_path):
    """every struct of the header, field by field: gcc compiles include/medfusion_hip.h as C (a cgo / JNI binding would) and prints sizeof and
    offsetof; the ctypes mirrors in medfusion_amd/lib.py must agree name by name"""
    import ctypes as C
    import shutil
    import subprocess
    from medfusion_amd import lib as L
    gcc = shutil.which("gcc")
    if gcc is None:
        pytest.skip("no gcc")
    structs = {"MfConvDesc": L.MfConvDesc, "MfSchedStep": L.MfSchedStep, "MfSchedArgs": L.MfSchedArgs, "MfGnFuse": L.MfGnFuse, "MfConvF16x2Call": L.MfConvF16x2Call,
               "MfWinoTail": L.MfWinoTail, "MfProfRow": L.MfProfRow}
    lines = ['#include <stdio.h>', '#include <stddef.h>', '#include "medfusion_hip.h"', 'int main(void) {']
    for name, cls in structs.items():
        lines.append(f'  printf("{name} %zu\\n", sizeof({name}));')
        for fname, _ in cls._fields_:
            lines.append(f'  printf("{name}.{fname} %zu\\n", offsetof({name}, {fname}));')
    lines += ['  return 0;', '}']
    src = tmp_path / "layout.c"
    src.write_text("\n".join(lines))
    exe = tmp_path / "layout"
    subprocess.run([gcc, "-std=c11", "-Wall", "-Werror", f"-I{ROOT / 'include'}", str(src), "-o", str(exe)], check=True)
    got = dict(ln.split() for ln in subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout.splitlines())
    for name, cls in structs.items():
        assert int(got[name]) == C.sizeof(cls), name
        for fname, _ in cls._fields_:
            assert int(got[f"{name}.{fname}"]) == getattr(cls, fname).offset, f"{name}.{fname}"


def test_the_documented_level2_binding_matches_the_header(tmp_path):
    """INTEGRATION.md's Level-2 ctypes stub, executed as written (up to its first function) against the built library and compared with the
    header: the version it asserts is MF_VERSION, its MfConvDesc lists the header's fields in the header's order, every mf_* it binds exists
    (the stub asserted ABI 220 while the library returned 230)"""
    import ctypes as C
    from medfusion_amd import lib as L
    md = (ROOT / "INTEGRATION.md").read_text()
    hdr = (ROOT / "include" / "medfusion_hip.h").read_text()
    block = re.search(r"## Level 2.*?```python\n(.*?)```", md, re.S).group(1)
    head = block.split("def conv2d_nhwc", 1)[0]
    assert "mf_version() ==" in head and "class MfConvDesc" in head
    L.load()
    head = head.replace('ctypes.CDLL("libmedfusion_hip.so")', f'ctypes.CDLL({str(L.LIB_PATH)!r})')
    ns = {}
    exec(compile(head, "INTEGRATION.md:level2", "exec"), ns)     # its own `assert _lib.mf_version() == N` runs here
    want = int(re.search(r"#define MF_VERSION (\d+)", hdr).group(1))
    assert f"mf_version() == {want}" in head
    body = re.search(r"typedef struct MfConvDesc \{(.*?)\} MfConvDesc;", hdr, re.S).group(1)
    body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
    body = re.sub(r"//[^\n]*", "", body)
    fields = [f.strip() for decl in re.findall(r"int32_t\s+([^;]+);", body) for f in decl.split(",")]
    assert [n for n, _ in ns["MfConvDesc"]._fields_] == fields == [n for n, _ in L.MfConvDesc._fields_]
    assert C.sizeof(ns["MfConvDesc"]) == C.sizeof(L.MfConvDesc)
    declared = set(re.findall(r"\b(mf_[a-z0-9_]+)\s*\(", hdr))
    live = re.sub(r"Removed \(.*?\n→", "→", md, flags=re.S)       # (the ABI history names entry points that were retired: not bindings)
    for name in set(re.findall(r"\b(mf_[a-z0-9_]+)\b", live)):
        ok = name in declared or (name.endswith("_") and any(d.startswith(name) for d in declared))   # `mf_cmdlist_*` style families
        assert ok, f"INTEGRATION.md names {name}, include/medfusion_hip.h does not declare it"


def test_host_validation_without_gpu():
    """Descriptor validation happens on the host before any launch: exercisable without a device."""
    import ctypes as C
    from medfusion_amd import kernels as K
    from medfusion_amd import lib as L
    lib = L.load()
    d = K.make_conv_desc(1, 4, 4, 32, 0, 32, 5, 1, 2)
    assert lib.mf_conv2d_workspace_bytes(C.byref(d)) == 0
    rc = lib.mf_conv2d_f32(None, None, None, None, None, None, 0, C.byref(d), None)
    assert rc == -2 and b"unsupported" in lib.mf_last_error()
    # split-K workspace sizing is a pure host computation: 8x8 level of the published UNet at B=16
    d = K.make_conv_desc(16, 8, 8, 1024, 1024, 1024, 3, 1, 1)
    need = lib.mf_conv2d_workspace_bytes(C.byref(d))
    assert need > 0 and need % (16 * 64 * 1024 * 4) == 0
    assert lib.mf_gn_stats_workspace_bytes(16, 1024, 256, 32) == 16 * 16 * 32 * 2 * 8


def test_product_never_imports_the_oracle():
    for py in (ROOT / "medfusion_amd").rglob("*.py"):
        tree = ast.parse(py.read_text())
        for node in ast.walk(tree):
            names = []
            if isinstance(node, ast.Import):
                names = [a.name for a in node.names]
            elif isinstance(node, ast.ImportFrom):
                names = [node.module or ""]
            for n in names:
                assert not n.split(".")[0] in ("oracle", "tests"), f"{py} imports {n}"
        assert "/root/reference" not in py.read_text()
    for src in (p for p in (ROOT / "medfusion_amd" / "csrc").iterdir() if p.is_file()):
        assert "oracle" not in src.read_text().lower().replace("oracle/synth.py: philox_normal is the spec", "")


def test_no_cpu_fallback():
    import medfusion_amd as M
    from medfusion_amd import kernels as K
    with pytest.raises(RuntimeError, match="no CPU"):
        K.gn_stats(torch.zeros((1, 2, 2, 32)), 8)
    u = M.UNet(in_ch=8, out_ch=8, spatial_dims=2, hid_chs=[32, 32, 64, 128], time_embedder_kwargs={"emb_dim": 64}, deep_supervision=False)
    with pytest.raises(RuntimeError, match="no CPU"):
        u(torch.zeros((1, 8, 8, 8)), torch.zeros((1,)))
    v = M.VAE(emb_channels=8, hid_chs=[32, 32, 64, 64])
    with pytest.raises(RuntimeError, match="no CPU"):
        v.decode(torch.zeros((1, 8, 4, 4)))
    p = M.DiffusionPipeline(M.GaussianNoiseScheduler, u, None, {"timesteps": 10})
    with pytest.raises(RuntimeError, match="no CPU"):
        p.sample(1, (8, 8, 8), steps=2)


def test_state_dict_keys_match_reference_layout():
    """Keys/shapes equal the oracle's, which gen_golden.py proved equal to the reference's."""
    import medfusion_amd as M
    from oracle import restate as R
    from tests.util import to_product_kwargs
    for att in ("none", "linear", "spatial"):
        kw = R.tiny_unet_kwargs(3, att, deep_supervision=True)
        a, b = M.UNet(**to_product_kwargs(kw)).state_dict(), R.UNet(**kw).state_dict()
        assert list(a) == list(b)
        assert all(a[k].shape == b[k].shape for k in a)
    a, b = M.VAE(**R.published_vae_kwargs(8)).state_dict(), R.VAE(**R.published_vae_kwargs(8)).state_dict()
    assert list(a) == list(b) and len(a) == 88
    a = M.GaussianNoiseScheduler(**R.published_scheduler_kwargs()).state_dict()
    b = R.GaussianNoiseScheduler(**R.published_scheduler_kwargs()).state_dict()
    assert list(a) == list(b) and all(torch.equal(a[k], b[k]) for k in a)
    pipe = M.DiffusionPipeline(M.GaussianNoiseScheduler, M.UNet, None, R.published_scheduler_kwargs(), to_product_kwargs(R.tiny_unet_kwargs(2)), use_ema=True)
    keys = list(pipe.state_dict())
    assert any(k.startswith("noise_estimator.in_conv.conv.weight") for k in keys)
    assert any(k.startswith("noise_scheduler.alphas_cumprod") for k in keys)
    assert any(k.startswith("ema_model.averaged_model.outc.conv.conv.weight") for k in keys)


def test_no_packed_fp32_instruction_takes_the_high_half_of_src1():
    """gfx950 erratum found in round 3 (scripts/pk_repro_min.hip, profiles/r03_pk_repro.txt): `v_pk_{mul,add,fma}_f32 ... op_sel:[x,1]` -- the LOW
    result reading the HIGH register of the src1 pair -- gets 0.0 for that operand in lanes 48..63 now and then (MFMAs + returning LDS reads
    on the other wave of the SIMD).  hipcc forms the selection by itself when it packs scalar fp32 code; it is what made the packed build of
    the fp16-pair convolution lose split-K partials.  No translation unit of the library may contain one (device assembly of every TU with
    its own flags, cached under csrc/build/lint), and conv_f16x2.hip contains no packed fp32 arithmetic at all."""
    from medfusion_amd import build as B
    assert B._PK_SRC1_HIGH.match("\tv_pk_mul_f32 v[2:3], v[2:3], v[34:35] op_sel:[0,1]")
    assert B._PK_SRC1_HIGH.match("\tv_pk_fma_f32 v[2:3], v[2:3], v[34:35], v[4:5] op_sel:[0,1,0]")
    assert B._PK_SRC1_HIGH.match("\tv_pk_add_f32 v[8:9], v[8:9], v[6:7] op_sel:[0,1] op_sel_hi:[1,0]")
    assert not B._PK_SRC1_HIGH.match("\tv_pk_fma_f32 v[2:3], v[34:35], v[2:3], v[4:5] op_sel:[1,0,0]")   # src0: measured clean
    assert not B._PK_SRC1_HIGH.match("\tv_pk_mul_f32 v[2:3], v[2:3], v[34:35] op_sel_hi:[1,0]")          # broadcast of the low half: clean
    assert B.lint_isa() == []
    asm = (B.OBJ / "lint" / "conv_f16x2.s").read_text()
    assert "v_pk_mul_f32" not in asm and "v_pk_add_f32" not in asm and "v_pk_fma_f32" not in asm


def test_command_list_host_contract():
    """mf_cmdlist_* without a GPU: recording is per thread and exclusive (begin twice -> error), end without begin -> error, an empty list
    has 0 launches and replays as a no-op, replay while recording is refused, free releases; every refusal leaves mf_last_error() set"""
    import ctypes
    import threading
    from medfusion_amd import lib as L
    lib = L.load()
    h = ctypes.c_void_p()
    assert lib.mf_cmdlist_end(ctypes.byref(h)) != 0 and b"not recording" in lib.mf_last_error()
    assert lib.mf_cmdlist_begin() == 0
    assert lib.mf_cmdlist_begin() != 0 and b"already recording" in lib.mf_last_error()
    other = {}

    def in_other_thread():      # another thread is NOT recording: it can begin / end its own list
        hh = ctypes.c_void_p()
        other["begin"] = lib.mf_cmdlist_begin()
        other["end"] = lib.mf_cmdlist_end(ctypes.byref(hh))
        other["count"] = lib.mf_cmdlist_count(hh)
        lib.mf_cmdlist_free(hh)
    t = threading.Thread(target=in_other_thread)
    t.start()
    t.join()
    assert other == {"begin": 0, "end": 0, "count": 0}
    empty = ctypes.c_void_p()
    assert lib.mf_cmdlist_end(ctypes.byref(empty)) == 0 and empty.value
    assert lib.mf_cmdlist_begin() == 0
    assert lib.mf_cmdlist_replay(empty, 1, None) != 0 and b"recording" in lib.mf_last_error()     # (this thread records again)
    assert lib.mf_cmdlist_end(ctypes.byref(h)) == 0
    assert lib.mf_cmdlist_count(h) == 0 and lib.mf_cmdlist_count(None) == 0
    assert lib.mf_cmdlist_replay(empty, 3, None) == 0        # nothing recorded: nothing launched (no device needed)
    assert lib.mf_cmdlist_replay(None, 1, None) != 0 and lib.mf_cmdlist_replay(empty, -1, None) != 0
    assert lib.mf_cmdlist_free(h) == 0 and lib.mf_cmdlist_free(empty) == 0


def test_a_plain_c_program_links_and_calls_the_library(tmp_path):
    """the boundary is a C ABI, not a ctypes convention: a C11 program that includes include/medfusion_hip.h links against
    libmedfusion_hip.so and calls host-side entry points (version, planner query, capability, error string) -- what a cgo / JNI stub would do"""
    import shutil
    import subprocess
    from medfusion_amd import build as B, lib as L
    gcc = shutil.which("gcc")
    if gcc is None:
        pytest.skip("no gcc")
    L.load()   # (built)
    src = tmp_path / "prog.c"
    src.write_text(r'''
#include <stdio.h>
#include <string.h>
#include "medfusion_hip.h"
int main(void) {
  MfConvDesc d;
  memset(&d, 0, sizeof d);
  d.N = 16; d.Hin = 32; d.Win = 32; d.C1 = 256; d.C2 = 0; d.Cout = 256; d.KH = 3; d.KW = 3; d.stride = 1; d.pad = 1;
  d.in_layout = MF_LAYOUT_NHWC; d.out_layout = MF_LAYOUT_NHWC; d.precision = MF_CONV_FP32_F16X2;
  int32_t tile = 0, sk = 0;
  int rc = mf_conv2d_plan_query(&d, &tile, &sk);
  printf("%d %d %d %d %d\n", mf_version(), rc, mf_conv2d_f16x2_ok(&d), (int)tile, (int)sk);
  d.KH = 5;
  printf("%d|%s\n", mf_conv2d_f16x2_ok(&d), mf_last_error());
  return 0;
}
''')
    exe = tmp_path / "prog"
    libdir = B.LIB.parent
    subprocess.run([gcc, "-std=c11", "-Wall", "-Werror", f"-I{ROOT / 'include'}", str(src), f"-L{libdir}", f"-l:{B.LIB.name}", f"-Wl,-rpath,{libdir}",
                    "-Wl,-rpath,/opt/rocm/lib", "-o", str(exe)], check=True)
    out = subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout.splitlines()
    ver, rc, ok, tile, sk = map(int, out[0].split())
    from medfusion_amd import kernels as K
    want = K.conv_plan(K.make_conv_desc(16, 32, 32, 256, 0, 256, 3, 1, 1, 0, precision=5))
    assert (ver, rc, ok) == (250, 0, 1) and (tile, sk) == want
    assert out[1].startswith("0|") and "unsupported" in out[1]


def test_host_side_under_address_sanitizer():
    """SURVEY section 5 (sanitizers row): the library's HOST side -- planner, descriptor validation, workspace arithmetic, the Winograd plan and
    table queries, struct passing through ctypes -- recompiled with -fsanitize=address (medfusion_amd.build.build_asan: host pass only, device
    code unchanged) and driven by the planner tests and the host-validation tests of this file in a subprocess that preloads the sanitizer's
    runtime: no report, same results."""
    import os
    import subprocess
    import sys
    from medfusion_amd import build as B
    try:
        rt = B.asan_runtime()
    except (subprocess.CalledProcessError, OSError, RuntimeError):
        pytest.skip("no clang AddressSanitizer runtime in this image")
    if not rt.exists():
        pytest.skip("no clang AddressSanitizer runtime in this image")
    lib = B.build_asan()
    env = dict(os.environ, LD_PRELOAD=str(rt), ASAN_OPTIONS="detect_leaks=0:abort_on_error=0", MEDFUSION_LIB=str(lib))
    r = subprocess.run([sys.executable, "-m", "pytest", "-q", "-x", "-p", "no:cacheprovider", "tests/test_planner_cpu.py",
                        "tests/test_boundary_cpu.py::test_host_validation_without_gpu", "tests/test_boundary_cpu.py::test_struct_layouts_match_header"],
                       cwd=str(ROOT), env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and "AddressSanitizer" not in r.stdout + r.stderr, (r.stdout[-3000:], r.stderr[-3000:])
    assert " passed" in r.stdout
