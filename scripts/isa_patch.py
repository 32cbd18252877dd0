#!/usr/bin/env python3
"""In-place ISA experiments on conv_f16x2.hip: compile once with -save-temps, edit the DEVICE assembly text, and re-run only the
assembler / device link / bundling / host embedding steps -- every other instruction of the translation unit keeps its place and its
encoding size, so a changed outcome belongs to the edited instructions alone (source-level hooks move hundreds of unrelated lines).

A packed fp32 VOP3P instruction is 8 bytes, its un-packed twin two 4-byte VOP2 instructions:
    v_pk_add_f32 v[a:a+1], v[b:b+1], v[c:c+1]                ->  v_add_f32_e32 va, vb, vc ; v_add_f32_e32 va+1, vb+1, vc+1
    v_pk_mul_f32 v[a:a+1], v[b:b+1], s[n:n+1] op_sel_hi:[1,0] ->  v_mul_f32_e32 va, sn, vb ; v_mul_f32_e32 va+1, sn, vb+1
so code size and every branch distance stay the same.

usage: isa_patch.py build     (here: medfusion_amd/csrc/build/variants/libmedfusion_hip_isa_<name>.so for every experiment below)
"""
import re
import shlex
import subprocess
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
WORK = ROOT / "medfusion_amd/csrc/build/isa"
DEV_S = "conv_f16x2-hip-amdgcn-amd-amdhsa-gfx950.s"

PK = re.compile(r"^\tv_pk_(add|mul)_f32 v\[(\d+):\d+\], ([vs])\[(\d+):\d+\], ([vs])\[(\d+):\d+\](.*)$")


def unpack(line):
    """the two-instruction twin of one packed add / mul, or None if the operand selection is one this script does not handle"""
    m = PK.match(line)
    if not m:
        return None
    op, d, k0, s0, k1, s1, mods = m.groups()
    d, s0, s1 = int(d), int(s0), int(s1)
    mods = mods.strip()
    # op_sel_hi:[x,y]: 0 = the HIGH result half takes the LOW register of that source (broadcast); default [1,1]
    hi0, hi1 = 1, 1
    mm = re.fullmatch(r"op_sel_hi:\[(\d),(\d)\]", mods) if mods else None
    if mods and not mm:
        return None
    if mm:
        hi0, hi1 = int(mm.group(1)), int(mm.group(2))
    lo = [(k0, s0), (k1, s1)]
    hi = [(k0, s0 + hi0), (k1, s1 + hi1)]

    def one(dst, a, b):
        if a[0] == "v" and b[0] == "s":   # VOP2: src0 may be an SGPR, vsrc1 must be a VGPR (both ops commute)
            a, b = b, a
        if b[0] == "s":
            return None
        return f"\tv_{op}_f32_e32 v{dst}, {a[0]}{a[1]}, {b[0]}{b[1]}"
    # in-place forms: the low instruction must not overwrite a register the high one still reads
    first, second = one(d, *lo), one(d + 1, *hi)
    if first is None or second is None:
        return None
    reads_hi = {f"{k}{n}" for k, n in hi}
    if f"v{d}" in reads_hi:
        first, second = second, first
        if f"v{d + 1}" in {f"{k}{n}" for k, n in lo}:
            return None
    return [first, second]


def kernels(text):
    """[(name, first line, last line)] of the conv kernels in the device assembly"""
    lines = text.split("\n")
    out, cur = [], None
    for i, ln in enumerate(lines):
        m = re.match(r"^(_ZN4mfc217conv_f16x2_kernel\w+):", ln)
        if m:
            cur = (m.group(1), i)
        if cur and ln.startswith(".Lfunc_end"):
            out.append((cur[0], cur[1], i))
            cur = None
    return lines, out


def patch(text, where):
    """un-pack the packed adds / muls of every conv kernel for which where(kind, lines, i) is true; kind in {"tree", "clobber", "final", "stage"}:
    prelude = the packed instructions between the last matrix instruction and the first hand-off store (value of this K slice)
    tree    = the adds behind the sc1 hand-off loads
    clobber = the packed instruction right behind a staging ds_write_b128 (it overwrites the store's first data pair)
    final   = the two packed multiplies right in front of a staging ds_write_b128 (they produce its data)
    stage   = every other packed add / mul between the first staging store and the end of the kernel"""
    lines, ks = kernels(text)
    n = 0
    for name, a, b in ks:
        i = a
        last_sc1_load = -1
        first_stage = None
        last_mfma = max((j for j in range(a, b) if "v_mfma" in lines[j]), default=a)
        first_sc1_store = next((j for j in range(a, b) if "buffer_store_dwordx4" in lines[j] and "sc1" in lines[j]), b)
        while i <= b:
            ln = lines[i]
            if "buffer_load_dwordx4" in ln and "sc1" in ln:
                last_sc1_load = i
            if "ds_write_b128" in ln and first_stage is None:
                first_stage = i
            if PK.match(ln):
                kind = None
                if last_mfma < i < first_sc1_store:
                    kind = "prelude"
                elif 0 <= last_sc1_load and i - last_sc1_load < 40 and first_stage is None:
                    kind = "tree"
                elif first_stage is not None or any("ds_write_b128" in lines[j] for j in range(i + 1, min(i + 4, b))):
                    prev = [l for l in lines[max(a, i - 3):i] if l.strip() and not l.strip().startswith(";")]
                    nxt = [l for l in lines[i + 1:i + 4] if l.strip() and not l.strip().startswith(";")]
                    if prev and "ds_write_b128" in prev[-1]:
                        kind = "clobber"
                    elif any("ds_write_b128" in l for l in nxt[:3]) and "v_pk_mul_f32" in ln:
                        kind = "final"
                    elif first_stage is not None:
                        kind = "stage"
                if kind and where(kind):
                    rep = unpack(ln)
                    if rep:
                        lines[i:i + 1] = rep
                        b += 1
                        i += 1
                        n += 1
            i += 1
    return "\n".join(lines), n


EXPERIMENTS = {
    "base": lambda kind: False,
    "tree": lambda kind: kind == "tree",
    "clobber": lambda kind: kind == "clobber",
    "final": lambda kind: kind == "final",
    "stage": lambda kind: kind in ("stage", "clobber", "final"),
    "prelude": lambda kind: kind == "prelude",
}


def commands():
    from medfusion_amd import build as B
    WORK.mkdir(parents=True, exist_ok=True)
    cmd = [B.hipcc(), *B.CFLAGS, "-c", str(B.CSRC / "conv_f16x2.hip"), "-o", "conv.o", "-save-temps", "-###"]   # packed fp32 ON
    out = subprocess.run(cmd, cwd=WORK, capture_output=True, text=True).stderr
    return [shlex.split(ln.strip()) for ln in out.splitlines() if ln.strip().startswith('"')]


def build():
    from medfusion_amd import build as B
    B.build(verbose=False)
    cmds = commands()
    for c in cmds:
        subprocess.run(c, cwd=WORK, check=True, capture_output=True)
    base = (WORK / DEV_S).read_text()
    first = next(i for i, c in enumerate(cmds) if "-cc1as" in c and DEV_S in c)
    vdir = B.OBJ / "variants"
    vdir.mkdir(exist_ok=True)
    for name, where in EXPERIMENTS.items():
        text, n = patch(base, where)
        (WORK / DEV_S).write_text(text)
        for c in cmds[first:]:
            subprocess.run(c, cwd=WORK, check=True, capture_output=True)
        obj = vdir / f"conv_f16x2_isa_{name}.o"
        (WORK / "conv.o").replace(obj)
        lib = vdir / f"libmedfusion_hip_isa_{name}.so"
        B._link([obj if s == "conv_f16x2.hip" else B.OBJ / (Path(s).stem + ".o") for s in B.SOURCES], lib, False)
        print(f"{name}: {n} packed instructions un-packed in place -> {lib}")
    (WORK / DEV_S).write_text(base)


if __name__ == "__main__":
    build()
