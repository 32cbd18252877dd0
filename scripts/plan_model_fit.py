#!/usr/bin/env python3
"""The planner's cost model (csrc/conv_f16x2.hip: make_plan2) replayed in Python over every conv sweep under profiles/ (and gpurun_out/), to tune it
WITHOUT a GPU: for every swept shape, the (tile, split-K) the model would pick, the time the sweep measured for that pick, and the regret against
the best of the sweep.  CPU only; `--model new|old`.  The numbers behind the round-6 model change are in profiles/r06_plan_model.txt."""
import argparse
import math
import re
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT / "scripts"))
sys.path.insert(0, str(ROOT))
from conv_sweep import unet_shapes, vae_shapes  # noqa: E402

# id: (BM, BN, WM, WN, NST, HG)
TILES = {31: (128, 256, 2, 4, 3, 0), 32: (256, 128, 4, 2, 3, 0), 33: (128, 128, 2, 4, 3, 0), 34: (128, 128, 4, 2, 3, 0), 35: (256, 64, 4, 2, 3, 0),
         36: (128, 64, 4, 2, 3, 0), 37: (64, 256, 1, 8, 3, 0), 51: (128, 128, 2, 2, 2, 0), 52: (128, 128, 2, 2, 3, 0), 53: (64, 128, 2, 2, 3, 0),
         54: (128, 64, 2, 2, 3, 0), 61: (256, 128, 4, 2, 3, 6), 62: (256, 128, 4, 2, 3, 7), 63: (128, 128, 2, 4, 3, 4), 64: (128, 128, 2, 4, 3, 5)}


def tile_lds(t):
    BM, BN, WM, WN, NST, HG = t
    return 2 * HG * WM * WN * 1024 + 3 * BN * 128 if HG else NST * (BM + BN) * 128


def percu(t):
    return 2 if tile_lds(t) <= 80 * 1024 else 1


def halo_fits(shape, t):
    N, H, W, Cin, Cout, k, st, ups = shape
    BM, BN, WM, WN, NST, HG = t
    if not HG:
        return True
    if k != 3 or st != 1 or ups:
        return False
    HW = H * W
    if HW >= BM:
        if HW % BM or BM % W:
            return False
        R, segs = BM // W, 1
    else:
        if BM % HW:
            return False
        R, segs = H, BM // HW
    return segs * (R + 2) * (W + 2) <= 8 * WM * WN * HG


def geometry(shape):
    N, H, W, Cin, Cout, k, st, ups = shape
    if ups == 3:                                     # the component GEMMs of a Winograd convolution: N = 16 x batch pseudo-samples of H W = T tile rows, 1x1
        return N * H * W, 1, Cin // 32
    if ups == 2:
        return N * 4 * H * W, 4, Cin // 32          # sub-pixel: M over (n, phase, y, x), 4 taps
    Ho = (H + 2 * (1 if k == 3 else 0) - k) // st + 1
    return N * Ho * Ho * (1 if H == W else 1), k * k, Cin // 32


def candidates(shape, halo_auto=False):
    N, H, W, Cin, Cout, k, st, ups = shape
    M, taps, cg = geometry(shape)
    for tid, t in TILES.items():
        BM, BN = t[0], t[1]
        if tid == 52 or (t[5] and not halo_auto):
            continue
        if Cout % BN or (ups == 2 and (H * W) % BM) or (ups == 3 and ((N // 16) * H * W) % BM) or not halo_fits(shape, t):
            continue
        s = 1
        while s <= 32 and s <= cg:
            chain_ok = -(-cg // s) * taps <= 96
            if chain_ok or not (s * 2 <= cg and s < 32):
                yield tid, s
            s *= 2


def cost_old(shape, tid, s):
    t = TILES[tid]
    BM, BN = t[0], t[1]
    M, taps, cg = geometry(shape)
    tiles = -(-M // BM) * (shape[4] // BN)
    pc = percu(t)
    wgs = tiles * s
    waves = -(-wgs // (256 * pc))
    its = -(-cg // s) * taps
    t_it = 0.68 * BM * BN / (128.0 * 128.0)
    if BM * BN <= 128 * 64:
        t_it = 0.80 if wgs > 256 else 0.43
    elif pc == 2:
        t_it = 1.40 if wgs > 256 else 0.75
    c = 7.0 + waves * (its + 3.0) * t_it
    if s > 1:
        c += 3.0 * int(math.log2(s))
    return c


# round 6 (csrc/conv_f16x2.hip, make_plan2, `fitted`): cost = 4.4 + W (its + 8) t_it(tile, dense) + 2.2 per tree level;  W = ceil(wgs / slots) up to two
# rounds, else wgs / slots + 0.24;  t_it per tile with one workgroup per CU (grid <= 256) / with the chip full -- least-squares fit (soft-L1 on
# log(model / measured)) to the ~2 900 (shape, tile, split-K, time) points of the sweeps listed in main()
TILE_COST = {31: (1.248, 1.332), 32: (1.306, 1.341), 33: (0.642, 0.677), 34: (0.621, 0.667), 35: (0.647, 0.762), 36: (0.404, 0.709), 37: (0.661, 0.835),
             51: (0.705, 1.225), 52: (0.656, 0.673), 53: (0.360, 0.675), 54: (0.377, 0.686), 61: (1.189, 1.215), 62: (1.192, 1.234), 63: (0.680, 0.685),
             64: (0.660, 0.691)}


# the component GEMMs (upsample == 3): the same form, constants from a regret-minimising search over profiles/r06_wino_tiles_b*.txt
TILE_COST_GEMM = {31: (1.160, 1.052), 32: (1.306, 1.341), 33: (0.596, 0.677), 34: (0.621, 0.667), 35: (0.647, 0.762), 36: (0.404, 0.709), 37: (0.630, 0.835),
                  51: (0.705, 1.225), 52: (0.656, 0.673), 53: (0.381, 0.660), 54: (0.377, 0.686)}


def cost_new(shape, tid, s, P=None):
    t = TILES[tid]
    BM, BN = t[0], t[1]
    M, taps, cg = geometry(shape)
    gemm = shape[7] == 3
    wgs = -(-M // BM) * (shape[4] // BN) * s
    slots = 256 * percu(t)
    its = -(-cg // s) * taps
    t_it = (TILE_COST_GEMM if gemm else TILE_COST)[tid][1 if wgs > 256 else 0]
    W = wgs / slots + 0.24 if wgs > 2 * slots else -(-wgs // slots)
    c = 4.4 + W * (its + (9.05 if gemm else 8.0)) * t_it
    if s > 1:
        c += (3.2 if gemm else 2.2) * int(math.log2(s))
    return c


P0 = None


def model_pick(shape, model="new"):
    """(tile, split-K) the planner's cost model picks for (N, H, W, Cin, Cout, k, stride, ups) -- the CPU replay of make_plan2 without its table"""
    if model == "new":
        cands = [c for c in candidates(shape, halo_auto=True) if c[0] not in (63, 64)]
        return min(cands, key=lambda c: cost_new(shape, c[0], c[1]))
    return min(candidates(shape), key=lambda c: cost_old(shape, c[0], c[1]))


# every sweep on file on the CURRENT kernels (path, batch, latent): what the model was fitted to and is checked against (tests/test_planner_cpu.py)
SWEEPS = [("profiles/r06_conv_sweep_b200.txt", 200, 32), ("profiles/r06_conv_sweep_b69.txt", 69, 32), ("profiles/r06_conv_sweep_b24.txt", 24, 32),
          ("profiles/r06_conv_sweep_b12.txt", 12, 32), ("profiles/r04_conv_sweep_b32.txt", 32, 32), ("profiles/r04_conv_sweep_b8.txt", 8, 32),
          ("profiles/r04_conv_sweep_planner_vs_best.txt", 16, 32), ("profiles/r04_conv_sweep_l64.txt", 8, 64), ("profiles/r04_conv_sweep_vae.txt", 16, 32)]


def regret(path, batch, latent, model="new"):
    """(time of the model's picks, time of the sweep's best picks) summed over the shapes of one sweep file, weighted by launches per UNet call;
    a pick outside the sweep's printed top 8 is priced at the 8th"""
    tp = tb = 0.0
    for nm, shape, cnt, t_auto, times in parse(ROOT / path, batch, latent):
        pick = model_pick(shape, model)
        tp += cnt * times.get(pick, max(max(times.values()), t_auto))
        tb += cnt * min(times.values())
    return tp, tb


# scripts/wino_sweep.py --tiles: GEMM + tail time of every Winograd-capable 3x3 by the component GEMM's (tile, split-K); "0/0" = the planner's own pick
WINO_TILE_SWEEPS = [(f"profiles/r06_wino_tiles_b{b}.txt", b, 32) for b in (8, 12, 16, 24, 32, 69, 200)]


def parse_wino_tiles(path, batch, latent=32):
    f = latent // 32
    dims = {nm: ((16 * n, 1, (h * f // 2) * (w * f // 2), c1 + c2, co, 1, 1, 3), cnt) for nm, n, h, w, c1, c2, co, k, st, ups, cnt in unet_shapes(batch) if k == 3 and st == 1 and not ups}
    rows, last = [], None
    for ln in Path(path).read_text().splitlines():
        m = re.match(r"(\S.*?\.c[01])\s+[\d.]+ \|", ln)
        if m:
            last = m.group(1).strip()
            continue
        if last and "GEMM + tail, us by (tile, split-K):" in ln:
            times = {(int(a), int(b)): float(c) for a, b, c in re.findall(r"(\d+)/(\d+):([\d.]+)", ln)}
            auto = times.pop((0, 0), None)
            shape, cnt = dims[last]
            rows.append((last, shape, cnt, auto if auto is not None else min(times.values()), times))
            last = None
    return rows


def gemm_model_pick(shape, model="new"):
    """the component GEMM's (tile, split-K) by the cost model: candidates as make_plan2 admits them for upsample == 3 (no halo tiles; the split-K must
    meet inside the launch: powers of two)"""
    cands = [c for c in candidates(shape) if TILES[c[0]][5] == 0]
    return min(cands, key=lambda c: (cost_new if model == "new" else cost_old)(shape, c[0], c[1]))


def parse(path, batch, latent=32, vae_batch=None):
    shapes = {}
    f = latent // 32
    for nm, n, h, w, c1, c2, co, k, st, ups, cnt in unet_shapes(batch) + vae_shapes(vae_batch or batch):
        shapes[nm] = ((n, h * f, w * f, c1 + c2, co, k, st, 2 if ups else 0), cnt)     # (conv_sweep.py --latent scales every shape, the VAE's too)
    rows = []
    for ln in Path(path).read_text().splitlines():
        m = re.match(r"(\S.*?)\s+(\d+)\s+(\d+)\s+(\d+)\s+([\d.]+) \|\s+([\d.]+)\s+[\d.]+ \| \((\d+),(\d+)\) ([\d.]+)\s+[\d.]+ \| (.*)$", ln)
        if not m or m.group(1).strip() not in shapes:
            continue
        shape, cnt = shapes[m.group(1).strip()]
        times = {(int(a), int(b)): float(c) for a, b, c in re.findall(r"(\d+)/(\d+):([\d.]+)", m.group(10))}
        rows.append((m.group(1).strip(), shape, cnt, float(m.group(6)), times))
    return rows


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", default="new")
    ap.add_argument("--verbose", action="store_true")
    a = ap.parse_args()
    for path, B, lat in SWEEPS:
        p = ROOT / path
        rows = parse(p, B, lat)
        tot_pick = tot_best = tot_auto = 0.0
        unknown = 0
        for nm, shape, cnt, t_auto, times in rows:
            best = min(times.values())
            pick = model_pick(shape, a.model)
            worst_listed = max(times.values())
            t_pick = times.get(pick)
            if t_pick is None:
                unknown += 1
                t_pick = max(worst_listed, t_auto) if a.model == "new" else t_auto
            tot_pick += cnt * t_pick
            tot_best += cnt * best
            tot_auto += cnt * t_auto
            if a.verbose and t_pick > 1.03 * best:
                print(f"   B={B} {nm:22s} pick {pick} {t_pick:.3f}  best {min(times, key=times.get)} {best:.3f}  (+{100 * (t_pick / best - 1):.1f} %)")
        print(f"{Path(path).name:40s} B={B:3d} lat={lat}: measured auto {tot_auto:7.3f} ms | model({a.model}) pick {tot_pick:7.3f} ms | best {tot_best:7.3f} ms | "
              f"pick / best {tot_pick / tot_best:.3f}  ({unknown} picks outside the sweep's top 8: priced at the 8th)")


def main_gemm(model):
    print("component GEMMs of the Winograd form (GEMM + tail, us per UNet evaluation; a pick outside the sweep's top 10 is priced at the 10th):")
    for path, B, lat in WINO_TILE_SWEEPS:
        if not (ROOT / path).exists():
            continue
        tp = tb = ta = 0.0
        miss = 0
        for nm, shape, cnt, t_auto, times in parse_wino_tiles(ROOT / path, B, lat):
            pick = gemm_model_pick(shape, model)
            if pick not in times:
                miss += 1
            tp += cnt * times.get(pick, max(max(times.values()), t_auto))
            tb += cnt * min(times.values())
            ta += cnt * t_auto
        print(f"{Path(path).name:40s} B={B:3d}: planner at sweep time {ta:8.1f} | model({model}) pick {tp:8.1f} | best {tb:8.1f} | pick / best {tp / tb:.3f} ({miss} outside the list)")


if __name__ == "__main__":
    main()
    main_gemm("old" if "--model" in sys.argv and "old" in sys.argv else "new")
