#!/usr/bin/env python3
"""Replays the rule of mf_wino_preferred (csrc/conv_f16x2_wino.inc: wino_rule) over every Winograd-vs-direct sweep on file (CPU only): per batch,
the time of the 3x3 stride-1 convolutions of one UNet evaluation on the direct form, with the per-shape best form, and with the rule's choice."""
import re
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT / "scripts"))
from conv_sweep import unet_shapes  # noqa: E402

# (file, batch, latent)
SWEEPS = [("profiles/r05_wino_sweep_b4.txt", 4, 32), ("profiles/r05_wino_sweep_b8.txt", 8, 32), ("profiles/r05_wino_sweep_b16.txt", 16, 32),
          ("profiles/r05_wino_sweep_b32.txt", 32, 32), ("profiles/r05_wino_sweep_l64.txt", 8, 64), ("profiles/r05_wino_sweep_b16_l64.txt", 16, 64),
          # round 6, on the round-6 planner, the component GEMM on the best (tile, split-K) of its search (what the round-6 GEMM model picks to 0 - 2.5 %)
          ("profiles/r06_wino_tiles_b8.txt", 8, 32), ("profiles/r06_wino_tiles_b12.txt", 12, 32), ("profiles/r06_wino_tiles_b16.txt", 16, 32),
          ("profiles/r06_wino_tiles_b24.txt", 24, 32), ("profiles/r06_wino_tiles_b32.txt", 32, 32), ("profiles/r06_wino_tiles_b69.txt", 69, 32),
          ("profiles/r06_wino_tiles_b200.txt", 200, 32)]


def wino_rule(n, h, w, cin, co):
    if cin * co < 190 * (cin + co):
        return False
    if h * w >= 1024 and co >= 512 and cin * co < 300 * (cin + co) and n * h * w >= 16384:
        return False
    return True


def rows(path, batch, latent):
    dims = {nm: (n, h * latent // 32, w * latent // 32, c1 + c2, co, cnt) for nm, n, h, w, c1, c2, co, k, st, ups, cnt in unet_shapes(batch) if k == 3 and st == 1 and not ups}
    out = []
    for ln in (ROOT / path).read_text().splitlines():
        m = re.match(r"(\S.*?\.c[01])\s+([\d.]+) \|\s+([\d.]+) \((\d+),(\d+)\) \+\s+([\d.]+) =\s+([\d.]+) \|\s+([\d.]+) \((\d+),(\d+)\)", ln)
        if m:
            out.append(dims[m.group(1).strip()] + (float(m.group(7)), float(m.group(8))))
    return out


def totals(path, batch, latent):
    d = b = r = 0.0
    for n, h, w, cin, co, cnt, td, tw in rows(path, batch, latent):
        d += cnt * td
        b += cnt * min(td, tw)
        r += cnt * (tw if wino_rule(n, h, w, cin, co) else td)
    return d, b, r


if __name__ == "__main__":
    for path, B, lat in SWEEPS:
        if not (ROOT / path).exists():
            continue
        d, b, r = totals(path, B, lat)
        print(f"{Path(path).name:34s} B = {B:3d} latent {lat}: direct {d:7.0f} us | per-shape best {b:7.0f} us | rule {r:7.0f} us  (rule / best {r / b:.3f}, rule vs direct {100 * (d / r - 1):+.1f} %)")
