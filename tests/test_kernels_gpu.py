"""Per-kernel parity on a real MI355X: each C-ABI entry point against a plain torch fp32 CPU reference of the same op
(floating-point kernels) or the oracle's bit-level spec (scheduler step, Philox)."""
import os
import sys
from pathlib import Path

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parents[1]

from oracle import restate as R
from oracle import synth as S
from tests.util import relerr


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "GPU tests need a ROCm device"
    import medfusion_amd  # noqa: F401  (loads libmedfusion_hip.so; raises if missing)
    return torch.device("cuda:0")


def _rand(name, shape, scale=1.0):
    return S.synth_input(name, shape, scale)


def _conv_ref(x, x2, w, b, stride, pad, ups):
    xin = x if x2 is None else torch.cat([x, x2], 1)
    if ups:
        xin = F.interpolate(xin, scale_factor=2, mode="nearest-exact")
    return F.conv2d(xin.double(), w.double(), b.double(), stride=stride, padding=pad).float()


CONV_CASES = [
    # (N, H, W, C1, C2, Cout, k, stride, ups)
    (2, 8, 8, 32, 0, 64, 3, 1, 0),
    (2, 8, 8, 64, 32, 32, 3, 1, 0),     # two-source (skip concat)
    (3, 10, 12, 32, 0, 32, 3, 2, 0),    # BasicDown stride 2, ragged M
    (2, 5, 6, 32, 0, 32, 3, 1, 1),      # BasicUp fused nearest x2
    (2, 8, 8, 64, 0, 128, 1, 1, 0),     # 1x1 conv_res
    (1, 16, 16, 256, 0, 256, 3, 1, 0),  # published 32^2-level shape (smaller HW)
    (2, 8, 8, 512, 512, 512, 3, 1, 0),  # out-block two-source, long K -> split-K
    (1, 7, 9, 96, 0, 96, 3, 1, 0),      # Cout = 96 (BN=32 path), ragged
]


IGEMM_BN = {1: 128, 2: 64, 3: 128, 4: 64, 5: 32, 6: 32, 7: 128, 8: 128, 9: 256, 10: 128, 23: 128, 24: 64, 27: 128, 28: 128}


def _igemm_tiles(case, tiles):
    """(case, tile) pairs that exist: the tile divides Cout and, for the BK = 64 forms, the channel counts are multiples of 64"""
    n, h, w, c1, c2, co, k, stride, ups = case
    return [t for t in tiles if t == 0 or (co % IGEMM_BN[t] == 0 and not (t in (23, 24, 27, 28) and (c1 % 64 or c2 % 64)))]


def _conv_operands(case, dev):
    from medfusion_amd import kernels as K
    n, h, w, c1, c2, co, k, stride, ups = case
    x = _rand(f"cx{case}", (n, c1, h, w))
    x2 = _rand(f"cy{case}", (n, c2, h, w)) if c2 else None
    wt = _rand(f"cw{case}", (co, c1 + c2, k, k), 1.0 / np.sqrt((c1 + c2) * k * k))
    b = _rand(f"cb{case}", (co,), 0.1)
    pad = R.monai_padding(k, stride)
    return x, x2, wt, b, pad, K.nchw_to_nhwc(x.to(dev)), (K.nchw_to_nhwc(x2.to(dev)) if c2 else None), K.pack_conv_weight(wt.to(dev))


@pytest.mark.parametrize("case,tile", [(c, t) for c in CONV_CASES for t in _igemm_tiles(c, [0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 23, 24, 27, 28])])
def test_conv_igemm(dev, case, tile):
    """MF_CONV_FP32 (v_mfma_f32_32x32x2_f32): an fp32 fma chain against an fp64 convolution"""
    from medfusion_amd import kernels as K
    n, h, w, c1, c2, co, k, stride, ups = case
    x, x2, wt, b, pad, xd, x2d, wp = _conv_operands(case, dev)
    want = _conv_ref(x, x2, wt, b, stride, pad, ups)
    for sk in ([0] if tile == 0 else [0, 1, 2, 3]):
        d = K.make_conv_desc(n, h, w, c1, c2, co, k, stride, pad, ups, tile_hint=tile, splitk_hint=sk)
        y = K.nhwc_to_nchw(K.conv2d(xd, wp, b.to(dev), d, x2=x2d))
        assert y.shape == want.shape
        assert relerr(y, want) < 1e-5, (case, tile, sk)  # fp32 fma chain vs fp64 reference, K up to 9216


@pytest.mark.parametrize("case,tile", [(c, t) for c in CONV_CASES for t in _igemm_tiles(c, [0, 1, 2, 3, 4, 6, 7, 8, 9, 10])])
def test_conv_igemm_split_bf16x3(dev, case, tile):
    """MF_CONV_FP32_SPLIT3_W3: fp32 operands split exactly into 3 bf16 terms (the weights once, mf_split_conv_weight_bf16x3), 6 product
    terms on the bf16 matrix cores, fp32 accumulation.  Same tolerance as the fp32-MFMA kernel (error measured against the fp64
    reference), and additionally the error must not exceed 3x the fp32 kernel's own error + 1e-6 (the same class, not merely 'within
    tolerance') whenever one accumulation chain is <= 96 chunks (what the planner guarantees when it is not overridden by a hint)."""
    from medfusion_amd import kernels as K
    n, h, w, c1, c2, co, k, stride, ups = case
    x, x2, wt, b, pad, xd, x2d, wp = _conv_operands(case, dev)
    want = _conv_ref(x, x2, wt, b, stride, pad, ups)
    w3 = K.split_conv_weight(wp)
    d0 = K.make_conv_desc(n, h, w, c1, c2, co, k, stride, pad, ups)
    e0 = relerr(K.nhwc_to_nchw(K.conv2d(xd, wp, b.to(dev), d0, x2=x2d)), want)
    chunks = k * k * (c1 + c2) // 32
    for sk in ([0] if tile == 0 else [0, 1, 2, 3]):
        d = K.make_conv_desc(n, h, w, c1, c2, co, k, stride, pad, ups, tile_hint=tile, splitk_hint=sk, precision=3)
        assert K.conv_is_igemm(d)
        e = relerr(K.nhwc_to_nchw(K.conv2d(xd, w3, b.to(dev), d, x2=x2d)), want)
        assert e < 1e-5, (case, tile, sk, e, e0)
        if sk == 0 or chunks / max(sk, 1) <= 96:
            assert e < 3 * e0 + 1e-6, (case, tile, sk, e, e0)


@pytest.mark.parametrize("case,tile", [(c, t) for c in CONV_CASES for t in _igemm_tiles(c, [0, 1, 4, 8, 9, 10])])
def test_conv_bf16_opt_in_mode(dev, case, tile):
    """MF_CONV_BF16 (opt-in, reduced precision): the kernel computes EXACTLY conv(bf16(x), bf16(w)) with fp32 accumulation (checked to
    1e-5 against an fp64 convolution of the rounded operands) -- and is therefore ~3e-3 away from the fp32 result (stated tolerance 2e-2)."""
    from medfusion_amd import kernels as K
    n, h, w, c1, c2, co, k, stride, ups = case
    x, x2, wt, b, pad, xd, x2d, wp = _conv_operands(case, dev)
    want32 = _conv_ref(x, x2, wt, b, stride, pad, ups)
    rb = lambda t: None if t is None else t.bfloat16().float()
    want16 = _conv_ref(rb(x), rb(x2), rb(wt), b, stride, pad, ups)
    wb = K.convert_conv_weight_bf16(wp)
    for sk in ([0] if tile == 0 else [0, 1, 3]):
        d = K.make_conv_desc(n, h, w, c1, c2, co, k, stride, pad, ups, tile_hint=tile, splitk_hint=sk, precision=4)
        y = K.nhwc_to_nchw(K.conv2d(xd, wb, b.to(dev), d, x2=x2d))
        assert relerr(y, want16) < 1e-5, (case, tile, sk)
        assert relerr(y, want32) < 2e-2, (case, tile, sk)


def test_conv_split_bf16x3_wide_dynamic_range(dev):
    """operands spanning 12 orders of magnitude (and exact zeros): the 3-way split is exact at every exponent, so the error
    relative to the fp64 result stays fp32-class per output element's own scale"""
    from medfusion_amd import kernels as K
    n, h, w, c, co = 1, 8, 8, 64, 64
    g = torch.Generator().manual_seed(5)
    x = torch.randn((n, c, h, w), generator=g) * torch.pow(10.0, torch.randint(-6, 7, (n, c, 1, 1), generator=g).float())
    x[:, ::7] = 0.0
    wt = torch.randn((co, c, 3, 3), generator=g) * 0.05
    b = torch.zeros(co)
    want = _conv_ref(x, None, wt, b, 1, 1, 0)
    scale = F.conv2d(x.double().abs(), wt.double().abs(), None, padding=1).float()   # per-element magnitude of the summands
    xd, wp = K.nchw_to_nhwc(x.to(dev)), K.pack_conv_weight(wt.to(dev))
    for prec in (0, 3, 5):
        d = K.make_conv_desc(n, h, w, c, 0, co, 3, 1, 1, 0, precision=prec)
        if prec == 5:
            y = K.nhwc_to_nchw(K.conv2d_f16x2(xd, K.split_weight_f16x2(wp), b.to(dev), d)).cpu()
        else:
            y = K.nhwc_to_nchw(K.conv2d(xd, K.split_conv_weight(wp) if prec == 3 else wp, b.to(dev), d)).cpu()
        # (the fp16-pair form scales per SAMPLE, not per channel: elements 12 decades below the sample's largest keep an absolute
        #  accuracy of 2^-50 of that maximum -- still far below the fp32 rounding of the sums they enter)
        assert float(((y - want).abs() / scale).max()) < 2e-6, prec


@pytest.mark.parametrize("case", [
    (2, 8, 8, 8, 32, 3, 1, "nchw", "nhwc"),    # in_conv: NCHW latent in
    (2, 8, 8, 32, 8, 1, 1, "nhwc", "nchw"),    # outc: NCHW out
    (2, 9, 7, 3, 16, 3, 1, "nchw", "nhwc"),    # VAE inc (Cin=3)
    (2, 8, 8, 64, 3, 1, 1, "nhwc", "nchw"),    # VAE outc
    (2, 8, 8, 64, 16, 3, 1, "nhwc", "nhwc"),   # out_enc
    (2, 8, 8, 8, 32, 1, 1, "nchw", "nhwc"),    # inc_dec conv_res
    (2, 8, 8, 16, 16, 1, 1, "nhwc", "nchw"),
    (2, 10, 12, 8, 16, 3, 2, "nhwc", "nhwc"),
    (2, 8, 8, 8, 64, 3, 1, "nchw", "nhwc"),     # small-Cin kernel (UNet in_conv shape family)
    (3, 9, 7, 8, 256, 3, 1, "nchw", "nhwc"),
    (1, 8, 8, 8, 512, 3, 1, "nchw", "nhwc"),    # VAE inc_dec 8->512 (two co tiles)
    (1, 8, 8, 8, 512, 1, 1, "nchw", "nhwc"),    # its 1x1 conv_res
    (2, 9, 7, 3, 64, 3, 1, "nchw", "nhwc"),     # VAE inc 3->64
    (2, 6, 6, 16, 128, 3, 2, "nhwc", "nhwc"),
    (3, 9, 7, 256, 8, 1, 1, "nhwc", "nchw"),    # conv_out1x1_kernel (UNet outc 256 -> 8): ragged pixel count, 8 lanes per pixel
    (2, 8, 8, 128, 5, 1, 1, "nhwc", "nhwc"),    # 4 lanes per pixel, Cout not a multiple of them, NHWC out
    (1, 5, 5, 1024, 2, 1, 1, "nhwc", "nchw"),   # 32 lanes per pixel
    (2, 8, 8, 96, 4, 1, 1, "nhwc", "nchw"),     # C / 32 not a power of two: the generic kernel
    (2, 8, 8, 32, 3, 1, 1, "nhwc", "nchw"),     # conv_out1x1_kernel with ONE lane per pixel, Cout = 3, NCHW out
    (1, 5, 5, 1024, 5, 1, 1, "nhwc", "nchw"),   # 32 lanes per pixel, Cout = 5, NCHW out
    (3, 7, 5, 32, 5, 1, 1, "nhwc", "nhwc"),     # one lane per pixel, Cout = 5, NHWC out, ragged pixel count
])
def test_conv_direct(dev, case):
    from medfusion_amd import kernels as K
    from medfusion_amd import lib as L
    n, h, w, ci, co, k, stride, lin, lout = case
    x = _rand(f"dx{case}", (n, ci, h, w))
    wt = _rand(f"dw{case}", (co, ci, k, k), 1.0 / np.sqrt(ci * k * k))
    b = _rand(f"db{case}", (co,), 0.1)
    pad = R.monai_padding(k, stride)
    want = _conv_ref(x, None, wt, b, stride, pad, 0)
    xd = x.to(dev) if lin == "nchw" else K.nchw_to_nhwc(x.to(dev))
    d = K.make_conv_desc(n, h, w, ci, 0, co, k, stride, pad, 0, L.LAYOUT_NCHW if lin == "nchw" else L.LAYOUT_NHWC,
                         L.LAYOUT_NCHW if lout == "nchw" else L.LAYOUT_NHWC)
    y = K.conv2d(xd, K.pack_conv_weight(wt.to(dev)), b.to(dev), d)
    if lout == "nhwc":
        y = K.nhwc_to_nchw(y)
    assert relerr(y, want) < 2e-6


def test_conv_direct_two_source(dev):
    from medfusion_amd import kernels as K
    from medfusion_amd import lib as L
    x, x2 = _rand("d2x", (2, 8, 6, 6)), _rand("d2y", (2, 8, 6, 6))
    wt, b = _rand("d2w", (4, 16, 1, 1), 0.25), _rand("d2b", (4,), 0.1)
    want = _conv_ref(x, x2, wt, b, 1, 0, 0)
    d = K.make_conv_desc(2, 6, 6, 8, 8, 4, 1, 1, 0, 0, L.LAYOUT_NHWC, L.LAYOUT_NCHW)
    y = K.conv2d(K.nchw_to_nhwc(x.to(dev)), K.pack_conv_weight(wt.to(dev)), b.to(dev), d, x2=K.nchw_to_nhwc(x2.to(dev)))
    assert relerr(y, want) < 2e-6


@pytest.mark.parametrize("case", [(2, 8, 8, 64, 0, 128, 0), (1, 16, 16, 128, 0, 128, 8), (3, 8, 16, 32, 32, 64, 4), (16, 8, 8, 512, 0, 512, 0), (2, 32, 32, 128, 0, 64, 0)])
def test_conv_subpixel_upsample(dev, case):
    """nearest-x2 + 3x3 conv (conv_blocks.py:123-125) in its sub-pixel form (upsample = 2) == the reference op; incl. split-K and
    the fused GroupNorm statistics, and equal (to rounding) to the gather form (upsample = 1)."""
    from medfusion_amd import kernels as K
    n, h, w, c1, c2, co, tile = case
    x = _rand(f"ux{case}", (n, c1, h, w))
    x2 = _rand(f"uy{case}", (n, c2, h, w)) if c2 else None
    wt = _rand(f"uw{case}", (co, c1 + c2, 3, 3), 1.0 / np.sqrt((c1 + c2) * 9))
    b = _rand(f"ub{case}", (co,), 0.1)
    want = _conv_ref(x, x2, wt, b, 1, 1, 1)
    xd = K.nchw_to_nhwc(x.to(dev))
    x2d = K.nchw_to_nhwc(x2.to(dev)) if c2 else None
    wsub = K.pack_upconv_weight(wt.to(dev))
    for sk in (0, 1, 2):
        d = K.make_conv_desc(n, h, w, c1, c2, co, 3, 1, 1, 2, tile_hint=tile, splitk_hint=sk)
        assert K.subpixel_ok(d)
        y = K.conv2d(xd, wsub, b.to(dev), d, x2=x2d)
        assert y.shape == (n, 2 * h, 2 * w, co)
        assert relerr(K.nhwc_to_nchw(y), want) < 1e-5, (case, sk)
        parts = K.conv_gn_parts(d, 8)
        if parts:
            y2, partial = K.conv2d_gn(xd, wsub, b.to(dev), d, 8, parts, x2=x2d)
            assert torch.equal(y2, y)
            assert relerr(partial[..., 0].sum(1), want.double().reshape(n, 8, -1).sum(-1)) < 1e-4
            assert relerr(partial[..., 1].sum(1), (want.double() ** 2).reshape(n, 8, -1).sum(-1)) < 1e-5
    d = K.make_conv_desc(n, 5, 6, c1, c2, co, 3, 1, 1, 2)  # 30 source pixels: not a multiple of 64 -> refused, gather form is used
    assert not K.subpixel_ok(d)


def test_conv_smallcin_two_source(dev):
    """in_conv with self-conditioning: torch.cat([x_t, self_cond]) as two NHWC sources of 8 channels each."""
    from medfusion_amd import kernels as K
    x, x2 = _rand("s2x", (2, 8, 6, 6)), _rand("s2y", (2, 8, 6, 6))
    wt, b = _rand("s2w", (64, 16, 3, 3), 1 / 12.0), _rand("s2b", (64,), 0.1)
    want = _conv_ref(x, x2, wt, b, 1, 1, 0)
    d = K.make_conv_desc(2, 6, 6, 8, 8, 64, 3, 1, 1)
    y = K.conv2d(K.nchw_to_nhwc(x.to(dev)), K.pack_conv_weight(wt.to(dev)), b.to(dev), d, x2=K.nchw_to_nhwc(x2.to(dev)))
    assert relerr(K.nhwc_to_nchw(y), want) < 2e-6


def test_conv_smallcin_round6_kernel_is_bit_identical_to_the_round5_one(dev, tmp_path):
    """conv_smallcin2_kernel (division-free patch fill, [k][pixel] patch) computes the same fma chain per output as conv_smallcin_kernel: the edge
    convolutions of the published models (UNet in_conv 8 -> 256 NCHW in, VAE inc 3 -> 64, VAE inc_dec 8 -> 512, the two-source self-conditioning form,
    a stride-2 case and a ragged last pixel group) give the SAME BITS from a process running the round-5 kernel (MF_SMALLCIN=0: read once per process)."""
    import os
    import subprocess
    import sys
    from pathlib import Path
    root = Path(__file__).resolve().parents[1]
    script = tmp_path / "smallcin.py"
    script.write_text(f"""
import sys
sys.path.insert(0, {str(root)!r})
import torch
from medfusion_amd import kernels as K, lib as L
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(1234)
outs = []
for n, h, w, c1, c2, co, k, st, nchw in [(16, 32, 32, 8, 0, 256, 3, 1, True), (2, 32, 32, 8, 0, 512, 3, 1, True), (2, 64, 64, 3, 0, 64, 3, 1, True),
                                         (2, 6, 6, 8, 8, 64, 3, 1, False), (3, 9, 7, 8, 0, 64, 3, 2, False), (1, 5, 5, 4, 0, 128, 1, 1, False)]:
    x = torch.randn((n, c1, h, w) if nchw else (n, h, w, c1), generator=g).to(dev)
    x2 = torch.randn((n, h, w, c2), generator=g).to(dev) if c2 else None
    wt = (torch.randn((co, c1 + c2, k, k), generator=g) * 0.2).to(dev)
    b = torch.randn((co,), generator=g).to(dev)
    d = K.make_conv_desc(n, h, w, c1, c2, co, k, st, 1 if k == 3 else 0, 0, in_layout=L.LAYOUT_NCHW if nchw else L.LAYOUT_NHWC)
    outs.append(K.conv2d(x, K.pack_conv_weight(wt), b, d, x2=x2).cpu())
torch.save(outs, sys.argv[1])
""")
    res = {}
    for mode in ("0", "1"):
        out = tmp_path / f"y{mode}.pt"
        r = subprocess.run([sys.executable, str(script), str(out)], env=dict(os.environ, MF_SMALLCIN=mode), capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        res[mode] = torch.load(out)
    assert len(res["0"]) == 6
    for a, b in zip(res["0"], res["1"]):
        assert a.shape == b.shape and torch.equal(a, b)
        assert bool(torch.isfinite(a).all()) and float(a.abs().max()) > 0


def test_conv_rejects_bad_descriptor(dev):
    from medfusion_amd import kernels as K
    x = torch.zeros((1, 4, 4, 32), device=dev)
    w = torch.zeros((32, 5, 5, 32), device=dev)
    d = K.make_conv_desc(1, 4, 4, 32, 0, 32, 5, 1, 2)
    with pytest.raises(RuntimeError, match="unsupported"):
        K.conv2d(x, w, None, d)


@pytest.mark.parametrize("case", [(2, 8, 8, 32, 32), (2, 8, 8, 64, 8), (3, 5, 7, 256, 32), (1, 64, 64, 64, 8), (2, 4, 4, 1024, 32), (1, 3, 3, 96, 8)])
def test_groupnorm_swish_residual_emb(dev, case):
    from medfusion_amd import kernels as K
    n, h, w, c, g = case
    x = _rand(f"gx{case}", (n, c, h, w), 2.0) + 0.7
    res = _rand(f"gr{case}", (n, c, h, w))
    emb = _rand(f"ge{case}", (n, c + 8))[:, 4:4 + c]
    gamma, beta = 1 + 0.2 * _rand(f"gg{case}", (c,)), _rand(f"gb{case}", (c,), 0.1)
    gn = F.group_norm(x.double(), g, gamma.double(), beta.double(), 1e-5)
    want = (gn * torch.sigmoid(gn) + res.double() + emb.double()[:, :, None, None]).float()
    xd = K.nchw_to_nhwc(x.to(dev))
    stats = K.gn_stats(xd, g)
    mean = x.double().reshape(n, g, -1).mean(-1)
    var = x.double().reshape(n, g, -1).var(-1, unbiased=False)
    assert relerr(stats[..., 0], mean.float()) < 1e-5
    assert relerr(stats[..., 1], (1 / torch.sqrt(var + 1e-5)).float()) < 1e-5
    embd = torch.zeros((n, c + 8), device=dev)
    embd[:, 4:4 + c] = emb.to(dev)
    e = embd[:, 4:4 + c]
    y = K.gn_apply(xd, stats, gamma.to(dev), beta.to(dev), g, 1, K.nchw_to_nhwc(res.to(dev)), e, e.stride(0))
    assert relerr(K.nhwc_to_nchw(y), want) < 3e-6
    # no-affine / no-act / in-place variants
    y2 = K.gn_apply(xd, stats, None, None, g, 0)
    assert relerr(K.nhwc_to_nchw(y2), F.group_norm(x.double(), g, None, None, 1e-5).float()) < 3e-6


@pytest.mark.parametrize("case", [(2, 16, 8, 64, 0, 128, 3, 32, 8), (16, 8, 8, 128, 128, 256, 3, 32, 0), (2, 32, 32, 64, 0, 64, 3, 8, 4), (1, 16, 16, 128, 0, 128, 1, 8, 8)])
def test_conv_fused_groupnorm_statistics(dev, case):
    """mf_conv2d_gn_f32 (stats from the conv epilogue or the split-K reducer) + mf_gn_finalize_f32 + mf_gn_apply_f32 == conv -> GroupNorm -> Swish + residual."""
    from medfusion_amd import kernels as K
    n, h, w, c1, c2, co, k, g, tile = case
    x = _rand(f"fx{case}", (n, c1, h, w))
    x2 = _rand(f"fy{case}", (n, c2, h, w)) if c2 else None
    wt = _rand(f"fw{case}", (co, c1 + c2, k, k), 1.0 / np.sqrt((c1 + c2) * k * k))
    b = _rand(f"fb{case}", (co,), 0.1) + 0.5
    res = _rand(f"fr{case}", (n, co, h, w))
    gamma, beta = 1 + 0.2 * _rand(f"fg{case}", (co,)), _rand(f"fz{case}", (co,), 0.1)
    pad = R.monai_padding(k, 1)
    yc = _conv_ref(x, x2, wt, b, 1, pad, 0).double()
    gn = F.group_norm(yc, g, gamma.double(), beta.double(), 1e-5)
    want = (gn * torch.sigmoid(gn) + res.double()).float()
    for sk in (0, 1, 2):
        d = K.make_conv_desc(n, h, w, c1, c2, co, k, 1, pad, 0, tile_hint=tile, splitk_hint=sk)
        parts = K.conv_gn_parts(d, g)
        xd = K.nchw_to_nhwc(x.to(dev))
        x2d = K.nchw_to_nhwc(x2.to(dev)) if c2 else None
        wp = K.pack_conv_weight(wt.to(dev))
        if parts > 0:
            y, partial = K.conv2d_gn(xd, wp, b.to(dev), d, g, parts, x2=x2d)
        else:
            y = K.conv2d(xd, wp, b.to(dev), d, x2=x2d)
            partial, parts = K.gn_stats_partial(y, g)
        ref_sum = yc.reshape(n, g, -1).sum(-1)
        assert relerr(partial[..., 0].sum(1), ref_sum) < 1e-5, (case, sk, parts)
        assert relerr(partial[..., 1].sum(1), (yc * yc).reshape(n, g, -1).sum(-1)) < 1e-5
        stats = K.gn_finalize(partial, parts, h * w, co, g)
        assert relerr(stats, K.gn_stats(y, g)) < 1e-6, (case, sk)
        out2 = K.gn_apply(y, stats, gamma.to(dev), beta.to(dev), g, 1, K.nchw_to_nhwc(res.to(dev)))
        assert relerr(K.nhwc_to_nchw(out2), want) < 1e-5, (case, sk, parts)
        # the same pass reducing the partial records itself (mf_gn_apply_from_partials_f32: no finalize launch)
        resd, records = K.nchw_to_nhwc(res.to(dev)), K.GnPartials(partial, parts, 1e-5)
        out3 = K.gn_apply(y, records, gamma.to(dev), beta.to(dev), g, 1, resd)
        assert relerr(out3, out2) < 2e-7, (case, sk, parts)
        bc = float(gamma.abs().max() * np.sqrt(co // g * h * w) + beta.abs().max())
        out4 = K.gn_apply(y, records, gamma.to(dev), beta.to(dev), g, 1, resd, split=True, bconst=bc)
        assert torch.equal(out4, out3) and torch.equal(out4._mf_split, K.split_f16x2(out3, out4._mf_bound))


def test_groupnorm_large_group_fp64_combine(dev):
    """VAE 256^2 level: 524 288 elements per group with a large mean -- fp32 E[x^2]-E[x]^2 would fail this."""
    from medfusion_amd import kernels as K
    x = torch.randn((1, 256, 256, 64), generator=torch.Generator().manual_seed(1)) * 0.5 + 30.0
    stats = K.gn_stats(x.to(dev), 8)
    xr = x.double().reshape(1, 256 * 256, 8, 8).permute(0, 2, 1, 3).reshape(1, 8, -1)
    assert relerr(stats[..., 0], xr.mean(-1).float()) < 1e-6
    assert relerr(stats[..., 1], (1 / torch.sqrt(xr.var(-1, unbiased=False) + 1e-5)).float()) < 2e-4


@pytest.mark.parametrize("case", [(4, 256, 1024, False, True), (16, 1024, 1024, False, False), (3, 64, 10240, True, False), (20, 48, 30, True, False), (1, 20, 7, False, False)])
def test_linear(dev, case):
    from medfusion_amd import kernels as K
    b, i, o, act_in, act_out = case
    x, w, bias = _rand(f"lx{case}", (b, i)), _rand(f"lw{case}", (o, i), 1 / np.sqrt(i)), _rand(f"lb{case}", (o,), 0.1)
    xin = x.double() * torch.sigmoid(x.double()) if act_in else x.double()
    want = xin @ w.double().T + bias.double()
    if act_out:
        want = want * torch.sigmoid(want)
    y = K.linear(x.to(dev), w.to(dev), bias.to(dev), act_in=act_in, act_out=act_out)
    assert relerr(y, want.float()) < 2e-6


def test_sinusoidal_and_embedding(dev):
    from medfusion_amd import kernels as K
    t = torch.tensor([0.0, 1.0, 37.0, 500.0, 999.0, 0.731])
    for dim, mp, flip in ((256, 10000, False), (20, 10, False), (16, 10000, True), (17, 10000, False)):
        want = R.SinusoidalPosEmb(dim, 1, mp, flip)(t)
        half = dim // 2
        freqs = torch.exp(-(np.log(mp) / (half - 1)) * torch.arange(half)).to(dev)  # host table, as the product passes it
        got = K.sinusoidal(t.to(dev), dim, float(mp), 1.0, flip, freqs=freqs)
        assert float((got.cpu() - want).abs().max()) < 5e-7, dim
        got = K.sinusoidal(t.to(dev), dim, float(mp), 1.0, flip)  # device expf fallback: 1 ulp in f_k times t <= 999
        assert float((got.cpu() - want).abs().max()) < 2e-4, dim
    table = _rand("emb_table", (3, 64))
    io = _rand("emb_io", (5, 64))
    idx = torch.tensor([2, 0, 1, 1, 2])
    got = K.embedding_add(table.to(dev), idx.to(dev), io.clone().to(dev))
    assert torch.equal(got.cpu(), io + table[idx])


def test_philox_normal_matches_spec(dev):
    from medfusion_amd import kernels as K
    out = torch.empty((6, 8, 16, 16), device=dev)
    K.philox_normal(out, seed=(7 << 32) | 123, draw=5, sample_offset=10)
    want = S.philox_normal((7 << 32) | 123, 5, np.arange(10, 16), 8 * 16 * 16).reshape(6, 8, 16, 16)
    assert float((out.cpu() - torch.from_numpy(want)).abs().max()) < 2e-6
    # shard invariance is exact: rows 12..15 alone
    part = torch.empty((4, 8, 16, 16), device=dev)
    K.philox_normal(part, seed=(7 << 32) | 123, draw=5, sample_offset=12)
    assert torch.equal(part, out[2:])
    # device-side step index
    step = torch.tensor([3], dtype=torch.int32, device=dev)
    ind = torch.empty_like(out)
    K.philox_normal(ind, seed=(7 << 32) | 123, draw=2, sample_offset=10, step_dev=step, draw_stride=1)
    assert torch.equal(ind, out)


@pytest.mark.parametrize("objective,clip", [("x_T", False), ("x_T", True), ("x_0", True), ("x_0", False)])
def test_sched_step_bit_exact(dev, objective, clip):
    """The fused step must be BIT-identical to the oracle's chain of fp32 elementwise ops (same inputs)."""
    import medfusion_amd as M
    from medfusion_amd import kernels as K
    from medfusion_amd import lib as L
    osch = R.GaussianNoiseScheduler(**R.published_scheduler_kwargs())
    psch = M.GaussianNoiseScheduler(**R.published_scheduler_kwargs())
    ts, _ = psch.loop_timesteps(7, True)
    recs = psch.step_records(ts, True)
    table = psch.upload_records(recs, dev)
    rev = list(reversed(ts))
    shape = (3, 8, 8, 8)
    for i in (0, 3, 6):
        t = torch.full((3,), rev[i], dtype=torch.long)
        x_t, pc, pu = _rand(f"sx{i}", shape), _rand(f"sp{i}", shape), _rand(f"su{i}", shape)
        npost, nddim = _rand(f"sn{i}", shape), _rand(f"sd{i}", shape)
        g = 8.0
        pred = pu + g * (pc - pu)
        osch.noise_fn = lambda like: npost
        if objective == "x_T":
            prior, x0 = osch.estimate_x_t_prior_from_x_T(x_t, t, pred, clip_x0=clip)
            xT = pred
        else:
            prior, x0 = osch.estimate_x_t_prior_from_x_0(x_t, t, pred, clip_x0=clip)
            xT = osch.estimate_x_T(x_t, x_0=pred, t=t, clip_x0=clip)
        want = prior
        if i < 6:  # DDIM update, diffusion_pipeline.py:297-304
            alpha, alpha_next = osch.alphas_cumprod[rev[i]], osch.alphas_cumprod[ts[7 - i - 2]]
            sigma = 1 * ((1 - alpha / alpha_next) * (1 - alpha_next) / (1 - alpha)).sqrt()
            c = (1 - alpha_next - sigma ** 2).sqrt()
            want = x0 * alpha_next.sqrt() + c * xT + sigma * nddim
        d = [v.to(dev) for v in (x_t, pc, pu, npost, nddim)]
        out, x0o, xTo = (torch.empty(shape, device=dev) for _ in range(3))
        a = L.MfSchedArgs(d[0].data_ptr(), d[1].data_ptr(), d[2].data_ptr(), None, d[3].data_ptr(), d[4].data_ptr(), 0, out.data_ptr(),
                          x0o.data_ptr(), xTo.data_ptr(), table.data_ptr(), None, i, 0 if objective == "x_T" else 1, int(clip), g, out.numel())
        K.sched_step(a)
        assert torch.equal(x0o.cpu(), x0), (i, "x0")
        assert torch.equal(xTo.cpu(), xT), (i, "xT")
        assert torch.equal(out.cpu(), want), (i, "x_t")


@pytest.mark.parametrize("objective,use_ddim,cfg", [("x_T", True, False), ("x_T", True, True), ("x_0", True, False), ("x_T", False, False)])
def test_loop_tail_in_one_launch_equals_the_four_launches(dev, objective, use_ddim, cfg):
    """mf_sched_step_philox_f32 (round 4: both noise draws in registers + the scheduler step + the step counter, one launch) against
    mf_philox_normal_f32 x 2 + mf_sched_step_f32 + mf_counter_add_i32 over a whole 7-iteration loop driven by the device counter: x_t and
    x_0 bit-identical at every iteration, the counter advances by one per launch, the ticket word returns to zero; a batch with a row offset
    (a shard of a multi-GPU batch) draws the rows of the global batch."""
    import medfusion_amd as M
    from medfusion_amd import kernels as K
    from medfusion_amd import lib as L
    psch = M.GaussianNoiseScheduler(**R.published_scheduler_kwargs())
    ts, steps = psch.loop_timesteps(7, use_ddim)
    table = psch.upload_records(psch.step_records(ts, use_ddim), dev)
    B, shape, seed, off, base = 5, (5, 8, 16, 24), 0x1234567890ABCDEF, 3, 1
    stride = 2 if use_ddim else 1
    xa = _rand("lt_x", shape).to(dev)
    xb = xa.clone()
    x0a, x0b = torch.empty_like(xa), torch.empty_like(xa)
    n_post, n_ddim = torch.empty_like(xa), torch.empty_like(xa)
    ca = torch.zeros(1, dtype=torch.int32, device=dev)
    cb = torch.zeros(2, dtype=torch.int32, device=dev)
    obj, g = 0 if objective == "x_T" else 1, 3.0
    for i in range(steps):
        pred = _rand(f"lt_p{i}", shape).to(dev)
        pu = _rand(f"lt_u{i}", shape).to(dev) if cfg else None
        # four launches
        K.philox_normal(n_post, seed, base, off, step_dev=ca, draw_stride=stride)
        if use_ddim:
            K.philox_normal(n_ddim, seed, base + 1, off, step_dev=ca, draw_stride=stride)
        a = L.MfSchedArgs(xa.data_ptr(), pred.data_ptr(), None if pu is None else pu.data_ptr(), None, n_post.data_ptr(), n_ddim.data_ptr() if use_ddim else None, 0,
                          xa.data_ptr(), x0a.data_ptr(), None, table.data_ptr(), ca.data_ptr(), 0, obj, 0, g, xa.numel())
        K.sched_step(a, outputs=(xa, x0a))
        K.counter_add(ca, 1)
        # one launch
        b = L.MfSchedArgs(xb.data_ptr(), pred.data_ptr(), None if pu is None else pu.data_ptr(), None, None, None, 0, xb.data_ptr(), x0b.data_ptr(), None,
                          table.data_ptr(), cb.data_ptr(), 0, obj, 0, g, xb.numel())
        K.sched_step_philox(b, seed, base, stride, off, B, cb, outputs=(xb, x0b))
        assert torch.equal(xa, xb) and torch.equal(x0a, x0b), (i, objective, use_ddim)
        assert cb.tolist() == [i + 1, 0] and int(ca.item()) == i + 1
    assert bool(xa.isfinite().all())


def test_sched_step_learned_variance(dev):
    import medfusion_amd as M
    from medfusion_amd import kernels as K
    from medfusion_amd import lib as L
    osch = R.GaussianNoiseScheduler(**R.published_scheduler_kwargs())
    psch = M.GaussianNoiseScheduler(**R.published_scheduler_kwargs())
    for tval in (0, 5, 999):
        recs = psch.step_records([tval], False)
        table = psch.upload_records(recs, dev)
        shape = (2, 8, 4, 4)
        t = torch.full((2,), tval, dtype=torch.long)
        x_t, pred, pv, npost = _rand("vx", shape), _rand("vp", shape), _rand("vv", shape, 0.5), _rand("vn", shape)
        osch.noise_fn = lambda like: npost
        prior, x0 = osch.estimate_x_t_prior_from_x_T(x_t, t, pred, clip_x0=False, var_scale=pv / 2 + 0.5)
        d = [v.to(dev) for v in (x_t, pred, pv, npost)]
        out, x0o = torch.empty(shape, device=dev), torch.empty(shape, device=dev)
        a = L.MfSchedArgs(d[0].data_ptr(), d[1].data_ptr(), None, d[2].data_ptr(), d[3].data_ptr(), None, 0, out.data_ptr(), x0o.data_ptr(), None,
                          table.data_ptr(), None, 0, 0, 0, 1.0, out.numel())
        K.sched_step(a)
        assert torch.equal(x0o.cpu(), x0)
        assert relerr(out, prior) < 1e-6  # expf on device vs CPU


@pytest.mark.parametrize("case", [(2, 4, 64, 64, 8), (2, 8, 100, 100, 32), (1, 3, 256, 256, 32), (2, 8, 64, 1, 4), (1, 8, 70, 130, 128), (2, 8, 1024, 1024, 32),
                                  (1, 8, 256, 256, 64), (3, 2, 33, 47, 16), (1, 8, 64, 64, 4), (2, 8, 16, 16, 8)])
def test_attention(dev, case):
    from medfusion_amd import kernels as K
    b, h, nq, nk, d = case
    c = h * d
    q, k, v = _rand(f"aq{case}", (b, c, nq)), _rand(f"ak{case}", (b, c, nk)), _rand(f"av{case}", (b, c, nk))
    want = R.compute_attention(q.double(), k.double(), v.double(), h, d ** -0.25).float()  # [B, C, Nq]
    tok = lambda z: z.transpose(1, 2).contiguous().to(dev)
    got = K.attention(tok(q), tok(k), tok(v), h, d ** -0.25)
    assert relerr(got.transpose(1, 2), want) < 5e-6


def test_layernorm_geglu_add_layout(dev):
    from medfusion_amd import kernels as K
    x = _rand("ln_x", (3, 5, 7, 96), 2.0)
    gmm, bta = 1 + 0.2 * _rand("ln_g", (96,)), _rand("ln_b", (96,), 0.1)
    want = F.layer_norm(x.double(), (96,), gmm.double(), bta.double(), 1e-5).float()
    assert relerr(K.layernorm(x.to(dev), gmm.to(dev), bta.to(dev)), want) < 3e-6
    h = _rand("geglu", (2, 4, 4, 64), 2.0)
    a, gate = h.double().chunk(2, dim=-1)
    assert relerr(K.geglu(h.to(dev)), (a * F.gelu(gate)).float()) < 3e-6
    y = _rand("lay", (2, 5, 6, 7))
    yd = y.to(dev)
    assert torch.equal(K.nhwc_to_nchw(K.nchw_to_nhwc(yd)), yd)
    assert torch.equal(K.nchw_to_nhwc(yd).cpu(), y.permute(0, 2, 3, 1).contiguous())
    assert torch.equal(K.add(yd, yd).cpu(), y + y)


def test_no_cpu_fallback():
    import medfusion_amd as M
    from medfusion_amd import kernels as K
    with pytest.raises(RuntimeError, match="no CPU"):
        K.gn_stats(torch.zeros((1, 2, 2, 32)), 8)
    u = M.UNet(**dict(in_ch=8, out_ch=8, spatial_dims=2, hid_chs=[32, 32, 64, 128], time_embedder_kwargs={"emb_dim": 64}, deep_supervision=False))
    with pytest.raises(RuntimeError, match="no CPU"):
        u(torch.zeros((1, 8, 8, 8)), torch.zeros((1,)))


def test_scheduler_tensor_api_per_row_t_bit_exact(dev):
    """SURVEY §8a row S2 with a DIFFERENT timestep per row (training / interpolate callers), bit-exact vs the oracle."""
    import medfusion_amd as M
    osch = R.GaussianNoiseScheduler(**R.published_scheduler_kwargs())
    psch = M.GaussianNoiseScheduler(**R.published_scheduler_kwargs()).to(dev)
    shape = (5, 8, 4, 4)
    x0, xT, xt, nz = (_rand(f"api{k}", shape) for k in range(4))
    t = torch.tensor([0, 17, 999, 500, 3])
    D = lambda z: z.to(dev)
    assert torch.equal(psch.estimate_x_t(D(x0), D(torch.tensor([-1, 0, 999, 1000, 42])), D(xT)).cpu(), osch.estimate_x_t(x0, torch.tensor([-1, 0, 999, 1000, 42]), xT))
    for clip in (True, False):
        assert torch.equal(psch.estimate_x_0(D(xt), D(xT), D(t), clip).cpu(), osch.estimate_x_0(xt, xT, t, clip))
        assert torch.equal(psch.estimate_x_T(D(xt), D(x0), D(t), clip).cpu(), osch.estimate_x_T(xt, x0, t, clip))
        osch.noise_fn = lambda like: nz
        for fn in ("estimate_x_t_prior_from_x_T", "estimate_x_t_prior_from_x_0"):
            pa, pb = getattr(psch, fn)(D(xt), D(t), D(xT), clip_x0=clip, noise=D(nz))
            oa, ob = getattr(osch, fn)(xt, t, xT, clip_x0=clip)
            assert torch.equal(pa.cpu(), oa) and torch.equal(pb.cpu(), ob), (fn, clip)
        tc = torch.tensor([5, 17, 999, 500, 3])  # cold diffusion uses t-1: keep t >= 1
        pa, pb = psch.estimate_x_t_prior_from_x_0(D(xt), D(tc), D(x0), clip_x0=clip, cold_diffusion=True)
        oa, ob = osch.estimate_x_t_prior_from_x_0(xt, tc, x0, clip_x0=clip, cold_diffusion=True)
        assert torch.equal(pa.cpu(), oa) and torch.equal(pb.cpu(), ob)
    assert torch.equal(psch.estimate_mean_t(D(xt), D(x0), D(t)).cpu(), osch.estimate_mean_t(xt, x0, t))
    assert torch.equal(psch.estimate_variance_t(D(t), 4).cpu(), osch.estimate_variance_t(t, 4))
    # scheduler_base.py:20-24 sample(): random t per row, x_T ~ N(0, 1), x_t from the forward process -- consistent with the pieces above
    x_t, x_T, ts = psch.sample(D(x0))
    assert ts.shape == (5,) and ts.dtype == torch.long and int(ts.min()) >= 0 and int(ts.max()) < psch.T
    assert x_T.shape == x0.shape and bool(x_T.isfinite().all()) and 0.5 < float(x_T.std()) < 1.5
    assert torch.equal(x_t.cpu(), osch.estimate_x_t(x0, ts.cpu(), x_T.cpu()))


def test_image_egress_uint8(dev):
    from medfusion_amd import kernels as K
    x = _rand("egress", (3, 3, 20, 24), 0.9)
    x[0, 0, 0, :4] = torch.tensor([-1.5, 1.5, 1.0, -1.0])
    got = K.image_to_uint8(x.to(dev)).cpu().numpy()
    img = x.numpy().clip(-1, 1)
    img = (img + 1) / 2 * 255                      # scripts/helpers/sample_dataset.py:46-47 verbatim arithmetic
    want = np.moveaxis(img, 1, -1).astype(np.uint8)
    assert np.array_equal(got, want)
    got1 = K.image_to_uint8(x.to(dev), normalize_each=True).cpu()
    r = ((x + 1) / 2).clamp(0, 1)                  # scripts/sample.py:49-50, then torchvision norm_ip + save_image
    r = torch.stack([(b.clamp(min=float(b.min()), max=float(b.max())) - b.min()) / max(float(b.max() - b.min()), 1e-5) for b in r])
    want1 = r.mul(255).add_(0.5).clamp_(0, 255).to(torch.uint8).permute(0, 2, 3, 1)
    assert torch.equal(got1, want1)


# ----------------------------------------------------------------------------- MF_CONV_FP32_F16X2 (fp16 pairs, LDS-DMA kernel)
def _decode_pairs(xs, shape):
    """int32 fp16-pair tensor -> float64 hi + lo / 2048 (without the per-sample scale)"""
    groups = shape[-1] // 8
    raw = xs.cpu().view(torch.int16).view(*shape[:-1], groups, 2, 8)            # [.., group, piece, 8] fp16 bit patterns
    hi = raw[..., 0, :].contiguous().view(torch.float16).double().reshape(shape)
    lo = raw[..., 1, :].contiguous().view(torch.float16).double().reshape(shape)
    return hi + lo / 2048.0


def test_split_f16x2_format(dev):
    """The fp16-pair form: groups of 8 channels [hi x 8][lo x 8] of x * 2^-s, s = floor(log2 bound[n]) - 14 per sample;
    (hi + lo / 2048) * 2^s reproduces x to one ulp of its fp32 value (2^-23 |x|: 23 of the 24 significand bits, exact for most values)
    at ANY magnitude -- rows of 1e9 and of 1e-9 alike -- and values far below the row's bound keep an absolute accuracy of 2^-50 bound."""
    from medfusion_amd import kernels as K
    x = _rand("splitx", (4, 5, 7, 64), 3.0)
    x[1] *= 1e9                                  # far beyond the fp16 range
    x[2] *= 1e-9                                 # far below it
    x[3, 0, 0, :8] = torch.tensor([0.0, -0.0, 1e-7, -3e-9, 1e-12, 2.0 ** -14, 5.0, -5.0])
    xd = x.to(dev)
    bound = K.maxabs_rows(xd)
    assert torch.equal(bound.cpu(), x.abs().amax(dim=(1, 2, 3)))
    xs = K.split_f16x2(xd, bound)
    s_exp = torch.floor(torch.log2(bound.cpu().double())) - 14
    back = _decode_pairs(xs, x.shape) * (2.0 ** s_exp).view(-1, 1, 1, 1)
    ref = x.double()
    err = (back - ref).abs()
    tol = ref.abs() * 2.0 ** -23 + (bound.cpu().double() * 2.0 ** -50).view(-1, 1, 1, 1)
    assert bool((err <= tol).all()), float((err / (ref.abs() + 1e-300)).max())
    assert float((err == 0).double().mean()) > 0.6                                 # three values out of four are exact
    big = ref.abs() > 1e-3 * bound.cpu().double().view(-1, 1, 1, 1)
    rms = float(((err[big] / ref.abs()[big]) ** 2).mean().sqrt())
    assert rms < 2.0 ** -24, rms                                                   # on average well below one fp32 rounding error
    # unscaled form (bound = None): identical for a row whose scale exponent is 0
    x1 = _rand("splitx1", (1, 4, 4, 32), 1.0) * 20000.0
    b1 = torch.full((1,), 30000.0)
    assert torch.equal(K.split_f16x2(x1.to(dev), b1.to(dev)), K.split_f16x2(x1.to(dev)))


F16X2_CASES = [
    # (N, H, W, C1, C2, Cout, k, stride, ups)
    (2, 8, 8, 32, 0, 64, 3, 1, 0),
    (2, 8, 8, 64, 32, 128, 3, 1, 0),     # two-source (skip concat)
    (3, 10, 12, 32, 0, 64, 3, 2, 0),     # BasicDown stride 2, ragged M
    (2, 8, 8, 32, 0, 256, 3, 1, 2),      # BasicUp, sub-pixel form (hw_src = 64: only the 64-row tile fits a phase)
    (2, 16, 16, 64, 0, 128, 3, 1, 2),    # BasicUp, sub-pixel form (hw_src = 256)
    (2, 8, 8, 64, 0, 128, 1, 1, 0),      # 1x1 conv_res
    (1, 16, 16, 256, 0, 256, 3, 1, 0),   # published 32^2-level shape (smaller HW)
    (2, 8, 8, 512, 512, 512, 3, 1, 0),   # out-block two-source, long K -> split-K
    (1, 7, 9, 96, 0, 192, 3, 1, 0),      # ragged M, Cout = 192
    # geometries of the halo tiles (61-64: whole image rows per tile, activations staged once per chunk)
    (2, 32, 32, 64, 0, 128, 3, 1, 0),    # 8 (tile 61) / 4 (tile 63) rows of a 32 x 32 image per tile
    (3, 16, 16, 64, 64, 128, 3, 1, 0),   # one whole 16 x 16 image per 256-row tile, two-source
    (5, 8, 8, 96, 32, 256, 3, 1, 0),     # four 8 x 8 images per 256-row tile (the last tile holds one), two per 128-row tile
    (1, 64, 64, 32, 0, 128, 3, 1, 0),    # 64-wide image: 4 / 2 rows per tile
]
F16X2_TILES = {31: (128, 256), 32: (256, 128), 33: (128, 128), 34: (128, 128), 35: (256, 64), 36: (128, 64), 37: (64, 256),
               51: (128, 128), 52: (128, 128), 53: (64, 128), 54: (128, 64),   # 51-54: 4-wave workgroups, two per CU
               61: (256, 128), 62: (256, 128), 63: (128, 128), 64: (128, 128)}   # 61-64: halo tiles (3x3 stride 1 only; HG = 6, 7, 4, 5)
HALO_HG = {61: (6, 8), 62: (7, 8), 63: (4, 8), 64: (5, 8)}   # tile -> (halo pieces per wave, waves)


def _halo_fits(case, tile):
    n, h, w, c1, c2, co, k, stride, ups = case
    if tile not in HALO_HG:
        return True
    bm, _ = F16X2_TILES[tile]
    if k != 3 or stride != 1 or ups:
        return False
    if h * w >= bm:
        if (h * w) % bm or bm % w:
            return False
        r, segs = bm // w, 1
    else:
        if bm % (h * w):
            return False
        r, segs = h, bm // (h * w)
    hg, nw = HALO_HG[tile]
    return segs * (r + 2) * (w + 2) <= 8 * nw * hg


def _f16x2_tiles(case):
    n, h, w, c1, c2, co, k, stride, ups = case
    return [0] + [t for t, (bm, bn) in F16X2_TILES.items() if co % bn == 0 and (ups != 2 or (h * w) % bm == 0) and _halo_fits(case, t)]


@pytest.mark.parametrize("case,tile", [(c, t) for c in F16X2_CASES for t in _f16x2_tiles(c)])
def test_conv_f16x2(dev, case, tile):
    """fp32 through pairs of fp16 (23-bit operands, 3 product terms, fp32 accumulate): error against an fp64 convolution within the
    fp32 tolerance AND of the same class as the fp32-MFMA kernel's own error; the fp16-pair mirror of the output equals the split of
    the output; the fused GroupNorm partial statistics equal the stand-alone pass."""
    from medfusion_amd import kernels as K
    n, h, w, c1, c2, co, k, stride, ups = case
    x = _rand(f"cx{case}", (n, c1, h, w))
    x2 = _rand(f"cy{case}", (n, c2, h, w)) if c2 else None
    wt = _rand(f"cw{case}", (co, c1 + c2, k, k), 1.0 / np.sqrt((c1 + c2) * k * k))
    b = _rand(f"cb{case}", (co,), 0.1)
    pad = R.monai_padding(k, stride)
    want = _conv_ref(x, x2, wt, b, stride, pad, 1 if ups else 0)
    xd = K.nchw_to_nhwc(x.to(dev))
    x2d = K.nchw_to_nhwc(x2.to(dev)) if c2 else None
    wp = K.pack_upconv_weight(wt.to(dev)) if ups == 2 else K.pack_conv_weight(wt.to(dev))
    wh = K.split_weight_f16x2(wp)
    d0 = K.make_conv_desc(n, h, w, c1, c2, co, k, stride, pad, ups)
    e0 = relerr(K.nhwc_to_nchw(K.conv2d(xd, wp, b.to(dev), d0, x2=x2d)), want)   # the fp32-MFMA kernel
    cgroups = (c1 + c2) // 32
    for sk in ([0] if tile == 0 else [0, 1, 2, 3, 4, 8]):   # 2, 4, 8: the slices meet inside the launch; 3: slabs + reducer pass
        if sk > cgroups:
            continue
        d = K.make_conv_desc(n, h, w, c1, c2, co, k, stride, pad, ups, tile_hint=tile, splitk_hint=sk, precision=5)
        assert K.conv_f16x2_ok(d), (case, tile, sk)
        y = K.conv2d_f16x2(xd, wh, b.to(dev), d, x2=x2d, measure_out=True)
        e = relerr(K.nhwc_to_nchw(y), want)
        assert e < 1e-5, (case, tile, sk, e, e0)
        assert e < 3 * e0 + 1e-6, (case, tile, sk, e, e0)
        if sk > 1:   # whichever workgroup of a pair arrives second, a + b is the same: bit-reproducible
            assert torch.equal(y, K.conv2d_f16x2(xd, wh, b.to(dev), d, x2=x2d)), (case, tile, sk)
        assert torch.equal(K.bound_of(y), y.abs().amax(dim=(1, 2, 3))), (case, tile, sk)     # the measured operand bound of the output
        # (measured by the convolution itself whenever a tile lies inside one sample, by a stand-alone pass otherwise)
        G = 8
        parts = K.conv_gn_parts(d, G)
        if parts:
            y2, partial = K.conv2d_f16x2(xd, wh, b.to(dev), d, x2=x2d, gn_groups=G, gn_parts=parts)
            assert torch.equal(y2, y)
            yg = y.double().cpu().reshape(n, -1, G, co // G)                                 # [N, HW, G, cpg]
            cnt = yg.shape[1] * yg.shape[3]
            mean, msq = yg.sum(dim=(1, 3)) / cnt, (yg * yg).sum(dim=(1, 3)) / cnt
            got = partial.sum(1).cpu() / cnt
            assert torch.allclose(got[..., 1], msq, rtol=1e-5, atol=0), (case, tile, sk)
            assert torch.allclose(got[..., 0], mean, rtol=0, atol=1e-5 * float(msq.max().sqrt())), (case, tile, sk)
    # operands of any magnitude: the same convolution on inputs scaled by 2^40 / 2^-40 per sample gives exactly the scaled result
    d = K.make_conv_desc(n, h, w, c1, c2, co, k, stride, pad, ups, tile_hint=tile, precision=5)
    y = K.conv2d_f16x2(xd, wh, None, d, x2=x2d)
    sc = torch.tensor([2.0 ** 40 if i % 2 == 0 else 2.0 ** -40 for i in range(n)], device=dev).view(-1, 1, 1, 1)
    ys = K.conv2d_f16x2(xd * sc, wh, None, d, x2=None if x2d is None else x2d * sc)
    assert torch.equal(ys, y * sc), (case, tile)
    if x2d is not None:   # the two sources of a fused concat carry their own scales
        yb = K.nhwc_to_nchw(K.conv2d_f16x2(xd * 1024.0, wh, b.to(dev), d, x2=x2d * (1.0 / 4096.0)))
        wantb = _conv_ref(x * 1024.0, x2 / 4096.0, wt, b, stride, pad, 1 if ups else 0)
        assert relerr(yb, wantb) < 1e-5, (case, tile)


@pytest.mark.parametrize("case", [(2, 16, 16, 256, 0, 256, 3), (2, 8, 8, 512, 512, 512, 3), (2, 16, 16, 512, 0, 256, 1), (1, 8, 8, 1024, 0, 1024, 3)])
def test_conv_f16x2_cancelling_sums(dev, case):
    """The 23-bit operand error of the fp16-pair arithmetic is relative to sum |w||x|, not to |y|: a convolution whose
    terms CANCEL -- activations 1 + 1e-3 noise, every filter with zero mean over its taps and input channels, so |y| ~ 1e-3 sum |w||x| in the
    interior of the image -- is checked element by element against fp64: |y - y64| <= 2^-21 sum |w||x|, the rigorous bound of the format (2^-23 per operand + the
    dropped lo.lo term < 2^-22; the fp32 accumulation of the matrix cores stays below it on chains <= 96 chunks), on the planner's own plan,
    the dominant tile and in-launch split-K.  The measured multiple of 2^-22 is printed (VERDICT's figure), next to the fp32-MFMA kernel
    (bit-for-bit an fp32 fma chain) on the same inputs, and the pair arithmetic must not be worse than 4x that kernel."""
    from medfusion_amd import kernels as K
    n, h, w, c1, c2, co, k = case
    cin = c1 + c2
    xa = 1.0 + 1e-3 * _rand(f"zx{case}", (n, cin, h, w))
    wt = _rand(f"zw{case}", (co, cin, k, k), 1.0 / np.sqrt(cin * k * k))
    wt = wt - wt.mean(dim=(1, 2, 3), keepdim=True)
    pad = R.monai_padding(k, 1)
    y64 = F.conv2d(xa.double(), wt.double(), None, padding=pad)
    s_abs = F.conv2d(xa.double().abs(), wt.double().abs(), None, padding=pad)
    inner = (slice(None), slice(None), slice(1, h - 1), slice(1, w - 1)) if k == 3 else (slice(None),) * 4
    cancel = float((y64[inner].abs() / s_abs[inner]).median())
    assert cancel < 5e-3, cancel                     # the sums really cancel (|y| << sum |w||x|) away from the padded border
    x, x2 = (xa[:, :c1], xa[:, c1:]) if c2 else (xa, None)
    xd = K.nchw_to_nhwc(x.contiguous().to(dev))
    x2d = K.nchw_to_nhwc(x2.contiguous().to(dev)) if c2 else None
    wp = K.pack_conv_weight(wt.to(dev))
    wh = K.split_weight_f16x2(wp)
    bound = 2.0 ** -21
    d0 = K.make_conv_desc(n, h, w, c1, c2, co, k, 1, pad, 0)
    r0 = float(((K.nhwc_to_nchw(K.conv2d(xd, wp, None, d0, x2=x2d)).cpu().double() - y64).abs() / s_abs).max())
    worst = 0.0
    for tile, sk in [(0, 0), (52, 1), (52, 2), (53, 4), (33, 1)]:
        if sk > cin // 32 or co % 128:
            continue
        d = K.make_conv_desc(n, h, w, c1, c2, co, k, 1, pad, 0, tile_hint=tile, splitk_hint=sk, precision=5)
        assert K.conv_f16x2_ok(d), (case, tile, sk)
        y = K.nhwc_to_nchw(K.conv2d_f16x2(xd, wh, None, d, x2=x2d)).cpu().double()
        r = float(((y - y64).abs() / s_abs).max())
        worst = max(worst, r)
        assert r <= bound and r <= 4 * r0 + 2.0 ** -26, (case, tile, sk, r, bound, r0)
    rel_y = worst * float((s_abs[inner] / y64[inner].abs().clamp_min(1e-300)).median())
    print(f"[measured] cancelling sums {case}: median |y| / sum|w||x| = {cancel:.1e}; max |y - y64| / sum|w||x|: fp16 pairs {worst:.2e} "
          f"(= {worst * 2 ** 22:.2f} x 2^-22), fp32 MFMA {r0:.2e}; i.e. ~{rel_y:.1e} relative to a typical |y|")


@pytest.mark.parametrize("case", [(2, 8, 8, 64, 32, 64, 3, 1, 0), (1, 16, 16, 256, 0, 256, 3, 1, 0), (2, 8, 8, 512, 512, 512, 3, 1, 0), (2, 8, 8, 64, 0, 128, 1, 1, 0),
                                  (2, 8, 8, 32, 0, 256, 3, 1, 2), (3, 10, 12, 32, 0, 64, 3, 2, 0)])
def test_conv_f16_single_term(dev, case):
    """MF_CONV_F16 (opt-in REDUCED precision): the fp16-pair operands and the LDS-DMA kernel with ONE product term, i.e. the
    operands rounded to fp16 -- on every tile and split-K the fp16-pair mode takes: error vs fp64 within the mode's own tolerance (2^-11 per
    operand), bit-reproducible, statistics / bounds like the fp32-class mode, and exactly the convolution of the fp16-rounded operands."""
    from medfusion_amd import kernels as K
    n, h, w, c1, c2, co, k, stride, ups = case
    x = _rand(f"hx{case}", (n, c1, h, w))
    x2 = _rand(f"hy{case}", (n, c2, h, w)) if c2 else None
    wt = _rand(f"hw{case}", (co, c1 + c2, k, k), 1.0 / np.sqrt((c1 + c2) * k * k))
    b = _rand(f"hb{case}", (co,), 0.1)
    pad = R.monai_padding(k, stride)
    want = _conv_ref(x, x2, wt, b, stride, pad, 1 if ups else 0)
    xd = K.nchw_to_nhwc(x.to(dev))
    x2d = K.nchw_to_nhwc(x2.to(dev)) if c2 else None
    wp = K.pack_upconv_weight(wt.to(dev)) if ups == 2 else K.pack_conv_weight(wt.to(dev))
    wh = K.split_weight_f16x2(wp)
    first = None
    for tile in _f16x2_tiles(case):
        for sk in ([0] if tile == 0 else [1, 2, 4]):
            if sk > (c1 + c2) // 32:
                continue
            d = K.make_conv_desc(n, h, w, c1, c2, co, k, stride, pad, ups, tile_hint=tile, splitk_hint=sk, precision=6)
            assert K.conv_f16x2_ok(d), (case, tile, sk)
            y = K.conv2d_f16x2(xd, wh, b.to(dev), d, x2=x2d, measure_out=True)
            e = relerr(K.nhwc_to_nchw(y), want)
            assert 1e-6 < e < 2e-3, (case, tile, sk, e)      # reduced precision: clearly not the fp32 class, clearly inside its own
            assert torch.equal(y, K.conv2d_f16x2(xd, wh, b.to(dev), d, x2=x2d)), (case, tile, sk)
            assert torch.equal(K.bound_of(y), y.abs().amax(dim=(1, 2, 3))), (case, tile, sk)
            if sk == 1:   # one accumulation chain per element: the same bits on every tile (tile 0 / sk 0 is the planner's own choice: it may split K)
                first = y if first is None else first
                assert torch.equal(y, first), (case, tile)


def _halo_table_entries():
    import re
    out = []
    for ln in (ROOT / "medfusion_amd" / "csrc" / "conv_plan_table.inc").read_text().splitlines():
        m = re.match(r"\s*\{([^}]*)\},", ln)
        if m:
            v = [int(x) for x in m.group(1).split(",")]
            if v[8] in HALO_HG:
                out.append(tuple(v))
    return out


@pytest.mark.parametrize("entry", _halo_table_entries())
def test_halo_plans_of_the_table_equal_the_nine_copy_kernel(dev, entry):
    """every shape the planner table sends to the halo-tile kernel (round 3: the large 3x3 stride-1 convolutions of the bigger workloads):
    the planner's own launch equals, bit for bit, the 9-copy kernel's launch with the same split-K (one association of the K slices), with
    the GroupNorm records and with a second source where the shape is a skip concat of two equal halves"""
    from medfusion_amd import kernels as K
    n, h, w, cin, co, k, stride, ups, tile, sk = entry
    two = cin >= 512 and (cin // 2) % 32 == 0          # (the up-path shapes: x ++ skip)
    c1, c2 = (cin // 2, cin // 2) if two else (cin, 0)
    g = torch.Generator().manual_seed(cin + co + n)
    x1 = torch.randn((n, h, w, c1), generator=g).to(dev)
    x2 = torch.randn((n, h, w, c2), generator=g).to(dev) * 3.0 if c2 else None
    wt = (torch.randn((co, k, k, cin), generator=g) * 0.02).to(dev)
    b = torch.randn((co,), generator=g).to(dev)
    wh = K.split_weight_f16x2(wt)
    d_auto = K.make_conv_desc(n, h, w, c1, c2, co, k, stride, 1, ups, precision=5)
    assert K.conv_plan(d_auto) == (tile, sk), (entry, K.conv_plan(d_auto))
    other = 52 if co % 128 == 0 else 53
    d_ref = K.make_conv_desc(n, h, w, c1, c2, co, k, stride, 1, ups, tile_hint=other, splitk_hint=sk, precision=5)
    assert K.conv_f16x2_ok(d_ref)
    G = 32
    ya, pa = K.conv2d_f16x2(x1, wh, b, d_auto, x2=x2, gn_groups=G, gn_parts=K.conv_gn_parts(d_auto, G))
    yr, pr = K.conv2d_f16x2(x1, wh, b, d_ref, x2=x2, gn_groups=G, gn_parts=K.conv_gn_parts(d_ref, G))
    assert torch.equal(ya, yr), entry
    assert relerr(pa.sum(1), pr.sum(1)) < 1e-5, entry      # (the records differ in number and in the fp32 grouping of their per-lane sums)


def test_split_from_slots_equals_finalize_then_split(dev):
    """mf_split_f16x2_slots (the fp16-pair mirror of a tensor whose bound still lies as slot maxima -- the slots a measuring convolution
    left, or those of mf_maxabs_rows_f32 with bound = NULL -- reduced inside the split) == mf_bound_finalize_f32 / mf_maxabs_rows_f32
    followed by mf_split_f16x2, bit for bit, and it publishes the same bound"""
    from medfusion_amd import kernels as K
    n, h, w, ci, co = 3, 16, 16, 64, 128
    x = K.nchw_to_nhwc(_rand("ss_x", (n, ci, h, w)).to(dev) * torch.tensor([1.0, 300.0, 1e-3], device=dev).view(n, 1, 1, 1))
    w1, b1 = _rand("ss_w", (co, ci, 1, 1), 0.2).to(dev), _rand("ss_b", (co,), 0.1).to(dev)
    d1 = K.make_conv_desc(n, h, w, ci, 0, co, 1, 1, 0, 0, precision=5)
    y = K.conv2d_f16x2(x, K.split_weight_f16x2(K.pack_conv_weight(w1)), b1, d1, measure_out=True)
    assert getattr(y, "_mf_slots", None) is not None and getattr(y, "_mf_bound", None) is None
    want_b = y.abs().amax(dim=(1, 2, 3))
    s = K.split_of(y)                                    # slots reduced inside the split
    assert torch.equal(y._mf_bound, want_b) and torch.equal(s, K.split_f16x2(y, want_b))
    z = _rand("ss_z", (5, 7, 9, 40)).to(dev) * torch.tensor([1.0, 0.0, 2e4, 3e-5, 7.0], device=dev).view(5, 1, 1, 1)   # (a zero row too)
    s = K.split_of(z)                                    # no slots yet: one measuring launch (slots only) + the same split
    assert torch.equal(z._mf_bound, z.abs().amax(dim=(1, 2, 3))) and torch.equal(s, K.split_f16x2(z, K.maxabs_rows(z)))


def test_f16x2_operands_carry_nan_and_inf_and_never_go_stale(dev):
    """(a) the fp16-pair split must not turn NaN / Inf into a finite number (a clamp with fminf / fmaxf does): a non-finite
    activation gives a non-finite convolution output, like the fp32 reference's; (b) the mirrors cached on a tensor are dropped by EVERY
    writer -- a torch in-place op (version counter) and every wrapper of this library that writes into an existing tensor."""
    from medfusion_amd import kernels as K
    n, h, w, c, co = 2, 8, 8, 64, 64
    x = _rand("nan_x", (n, h, w, c)).to(dev)
    wt = K.split_weight_f16x2(K.pack_conv_weight(_rand("nan_w", (co, c, 3, 3), 0.05).to(dev)))
    d = K.make_conv_desc(n, h, w, c, 0, co, 3, 1, 1, 0, precision=5)
    y0 = K.conv2d_f16x2(x, wt, None, d)
    assert bool(torch.isfinite(y0).all())
    for bad in (float("nan"), float("inf"), float("-inf")):
        xb = x.clone()
        xb[1, 3, 4, 7] = bad
        s = K.split_f16x2(xb, xb.abs().nan_to_num(posinf=1e30).amax(dim=(1, 2, 3)))
        raw = s.view(torch.int32)[1, 3, 4, 0:8].view(torch.float16)       # the 8-channel group [hi x 8 | lo x 8] holding channel 7
        assert not bool(torch.isfinite(raw[7])), (bad, raw)
        yb = K.conv2d_f16x2(xb, wt, None, d)
        assert not bool(torch.isfinite(yb[1]).all()), bad                 # the sample with the bad value
        assert torch.equal(yb[0], y0[0]), bad                             # the other sample is untouched
    # (b) stale mirrors
    x1 = x.clone()
    s1, b1 = K.split_of(x1), K.bound_of(x1)
    x1.mul_(3.0)                                                          # torch in-place: the version counter moves
    assert K.stale(x1)
    s2, b2 = K.split_of(x1), K.bound_of(x1)
    assert torch.equal(b2, x1.abs().amax(dim=(1, 2, 3))) and torch.equal(s2, K.split_f16x2(x1, b2)) and not torch.equal(b1, b2)
    for write in (lambda t: K.rows_axpby(x, out=t), lambda t: K.philox_normal(t, 1, 0), lambda t: K.add(x, x, out=t)):
        t = x.clone()
        K.split_of(t)
        write(t)
        assert getattr(t, "_mf_split", None) is None and getattr(t, "_mf_bound", None) is None
        assert torch.equal(K.split_of(t), K.split_f16x2(t, t.abs().amax(dim=(1, 2, 3))))


def test_gn_apply_split_mirror(dev):
    """gn_apply(split=True) writes the fp16-pair mirror of exactly what it writes in fp32, scaled by the bound it derives and
    publishes: bconst + bound(residual) + bound(embedding row) >= max |out|"""
    from medfusion_amd import kernels as K
    x = _rand("gas_x", (2, 6, 6, 64)).to(dev)
    res = (_rand("gas_r", (2, 6, 6, 64)) * torch.tensor([1.0, 3e6]).view(2, 1, 1, 1)).to(dev)   # one sample far outside the fp16 range
    emb = _rand("gas_e", (2, 64)).to(dev)
    gamma, beta = _rand("gas_g", (64,)).to(dev), _rand("gas_b", (64,)).to(dev)
    stats = K.gn_stats(x, 8)
    bc = float(gamma.abs().max()) * (36 * 8) ** 0.5 + float(beta.abs().max())
    y = K.gn_apply(x, stats, gamma, beta, 8, 1, res, emb, emb.stride(0), split=True, bconst=bc)
    assert torch.equal(y, K.gn_apply(x, stats, gamma, beta, 8, 1, res, emb, emb.stride(0)))
    assert bool((y._mf_bound >= y.abs().amax(dim=(1, 2, 3))).all())
    want_b = bc + res.abs().amax(dim=(1, 2, 3)) + emb.abs().amax(dim=1)
    assert torch.allclose(y._mf_bound, want_b, rtol=1e-6)
    assert torch.equal(y._mf_split, K.split_f16x2(y, y._mf_bound))
    K.add(y, res, out=y)                      # writing into a tensor drops its (now stale) mirrors
    assert getattr(y, "_mf_split", None) is None and getattr(y, "_mf_bound", None) is None


def test_gn_apply_from_partials_with_residual_bound_slots(dev):
    """the block epilogue of a channel-changing ResBlock in the default arithmetic: 3x3 conv leaving GroupNorm partial records, 1x1 conv_res
    leaving per-(tile, wave) maxima; mf_gn_apply_from_partials_f32 reduces both itself and must equal the path through the finalize launches"""
    from medfusion_amd import kernels as K
    n, h, w, ci, co, g = 3, 16, 16, 64, 128, 8
    x = K.nchw_to_nhwc(_rand("rs_x", (n, ci, h, w)).to(dev) * torch.tensor([1.0, 300.0, 1e-3], device=dev).view(n, 1, 1, 1))
    w3, w1 = _rand("rs_w3", (co, ci, 3, 3), 0.05).to(dev), _rand("rs_w1", (co, ci, 1, 1), 0.2).to(dev)
    b3, b1 = _rand("rs_b3", (co,), 0.1).to(dev), _rand("rs_b1", (co,), 0.1).to(dev)
    gamma, beta = (1 + 0.2 * _rand("rs_g", (co,))).to(dev), _rand("rs_z", (co,), 0.1).to(dev)
    d3 = K.make_conv_desc(n, h, w, ci, 0, co, 3, 1, 1, 0, precision=5)
    d1 = K.make_conv_desc(n, h, w, ci, 0, co, 1, 1, 0, 0, precision=5)
    parts = K.conv_gn_parts(d3, g)
    assert parts > 0
    y, partial = K.conv2d_f16x2(x, K.split_weight_f16x2(K.pack_conv_weight(w3)), b3, d3, gn_groups=g, gn_parts=parts)
    res = K.conv2d_f16x2(x, K.split_weight_f16x2(K.pack_conv_weight(w1)), b1, d1, measure_out=True)
    assert getattr(res, "_mf_slots", None) is not None and getattr(res, "_mf_bound", None) is None
    bc = float(gamma.abs().max()) * (h * w * co // g) ** 0.5 + float(beta.abs().max())
    a = K.gn_apply(y, K.GnPartials(partial, parts, 1e-5), gamma, beta, g, 1, res, split=True, bconst=bc)      # slots reduced inside the pass
    assert getattr(res, "_mf_bound", None) is None
    want_bound = bc + res.abs().amax(dim=(1, 2, 3))
    assert torch.equal(a._mf_bound, want_bound)
    assert torch.equal(K.bound_of(res), res.abs().amax(dim=(1, 2, 3)))                                       # lazily finalised on demand
    b = K.gn_apply(y, K.gn_finalize(partial, parts, h * w, co, g), gamma, beta, g, 1, res, split=True, bconst=bc)
    assert relerr(a, b) < 2e-7 and torch.equal(a._mf_bound, b._mf_bound)
    assert torch.equal(a._mf_split, K.split_f16x2(a, a._mf_bound))


def test_an_understated_operand_bound_fails_loudly(dev):
    """the apply pass and the fused tails split WITHOUT the fp16 range clamp (their bound is derived, the clamp can never act).  What an
    UNDERSTATED bound does -- a caller of the C-ABI handing in a residual bound below the data -- is therefore defined here: the scaled value leaves
    the fp16 range, its pair becomes (Inf, NaN), and the convolution that consumes the tensor returns non-finite numbers for the samples concerned.
    Loud, not silently saturated (a clamp would hand a finite, wrong operand to the next layer); samples whose bound holds are untouched."""
    from medfusion_amd import kernels as K
    n, h, w, c, g = 2, 8, 8, 64, 8
    y = K.nchw_to_nhwc(_rand("ub_y", (n, c, h, w)).to(dev))
    res = K.nchw_to_nhwc(_rand("ub_r", (n, c, h, w), 50.0).to(dev))
    partial, parts = K.gn_stats_partial(y, g)
    true_b = res.abs().amax(dim=(1, 2, 3))
    res._mf_bound = torch.stack([true_b[0], true_b[1] / 64.0])          # sample 1: understated by 2^6
    K._stamp(res)
    out = K.gn_apply(y, K.GnPartials(partial, parts, 1e-5), None, None, g, 1, res, split=True, bconst=1e-3)   # (bconst tiny: the residual bound carries the scale)
    assert bool(torch.isfinite(out).all())                               # the fp32 form is what it is
    raw = out._mf_split.view(torch.float16).reshape(n, -1).float()
    assert bool(torch.isfinite(raw[0]).all()) and not bool(torch.isfinite(raw[1]).all())
    wt = _rand("ub_w", (64, c, 3, 3), 0.05).to(dev)
    d = K.make_conv_desc(n, h, w, c, 0, 64, 3, 1, 1, 0, precision=5)
    z = K.conv2d_f16x2(out, K.split_weight_f16x2(K.pack_conv_weight(wt)), None, d)
    assert bool(torch.isfinite(z[0]).all()) and not bool(torch.isfinite(z[1]).all())


def test_gn_apply_pairs_only_output_and_residual_from_pairs(dev):
    """between the two convolutions of a ResBlock the activation exists as fp16 pairs only: the apply pass writes no fp32 (out_fp32=False)
    and the next pass reads its residual from the pairs -- the pairs are bit-identical to those of the ordinary pass, the residual read
    back is the fp32 value with at most its last significand bit cleared, and every fp32 reader refuses such a tensor"""
    from medfusion_amd import kernels as K
    n, h, w, ci, co, g = 3, 16, 16, 64, 128, 8
    x = K.nchw_to_nhwc(_rand("po_x", (n, ci, h, w)).to(dev))
    w3 = K.split_weight_f16x2(K.pack_conv_weight(_rand("po_w3", (co, ci, 3, 3), 0.05).to(dev)))
    w3b = K.split_weight_f16x2(K.pack_conv_weight(_rand("po_w3b", (co, co, 3, 3), 0.05).to(dev)))
    b3 = _rand("po_b3", (co,), 0.1).to(dev)
    gamma, beta = (1 + 0.2 * _rand("po_g", (co,))).to(dev), _rand("po_z", (co,), 0.1).to(dev)
    emb = _rand("po_e", (n, co)).to(dev)
    d3 = K.make_conv_desc(n, h, w, ci, 0, co, 3, 1, 1, 0, precision=5)
    d3b = K.make_conv_desc(n, h, w, co, 0, co, 3, 1, 1, 0, precision=5)
    parts = K.conv_gn_parts(d3, g)
    bc = float(gamma.abs().max()) * (h * w * co // g) ** 0.5 + float(beta.abs().max())

    def first(out_fp32):
        y, partial = K.conv2d_f16x2(x, w3, b3, d3, gn_groups=g, gn_parts=parts)
        return K.gn_apply(y, K.GnPartials(partial, parts, 1e-5), gamma, beta, g, 1, None, emb, emb.stride(0), out=y, split=True, bconst=bc, out_fp32=out_fp32)

    a_full, a_po = first(True), first(False)
    assert not K.pairs_only(a_full) and K.pairs_only(a_po)
    assert torch.equal(a_full._mf_split, a_po._mf_split) and torch.equal(a_full._mf_bound, a_po._mf_bound)
    for reader in (lambda t: K.nhwc_to_nchw(t), lambda t: K.add(t, t), lambda t: K.gn_stats(t, g), lambda t: K.conv2d(t, K.pack_conv_weight(_rand("po_w1", (co, co, 1, 1)).to(dev)), None,
                                                                                                                   K.make_conv_desc(n, h, w, co, 0, co, 1, 1, 0, 0))):
        with pytest.raises(RuntimeError, match="fp16 pairs"):
            reader(a_po)

    def second(a):   # the second block: conv on the pairs, identity residual = a
        y, partial = K.conv2d_f16x2(a, w3b, b3, d3b, gn_groups=g, gn_parts=K.conv_gn_parts(d3b, g))
        return K.gn_apply(y, K.GnPartials(partial, K.conv_gn_parts(d3b, g), 1e-5), gamma, beta, g, 1, a, None, 0, out=y, split=True, bconst=bc)

    o_full, o_po = second(a_full), second(a_po)
    assert not K.pairs_only(o_po)
    # the residual read from pairs = the fp32 residual with <= one ulp cleared: |difference| <= 2^-23 |residual| + rounding of the add
    diff = (o_full - o_po).abs()
    assert float((diff / (a_full.abs() * 2.0 ** -22 + o_full.abs() * 2.0 ** -23 + 1e-30)).max()) <= 1.0
    assert float(diff.max()) > 0 or True
    assert torch.equal(o_po._mf_bound, o_full._mf_bound)


@pytest.mark.parametrize("case", [(16, 32, 32, 256, 0, 256, 3, 2, 0, 0, 0), (16, 8, 8, 512, 0, 512, 3, 1, 2, 0, 0), (16, 16, 16, 512, 0, 512, 3, 2, 0, 53, 4),
                                  (3, 10, 12, 32, 0, 64, 3, 2, 0, 0, 0), (2, 8, 8, 64, 32, 128, 3, 1, 0, 0, 0), (2, 16, 16, 64, 0, 128, 3, 1, 2, 0, 0),
                                  (16, 8, 8, 1024, 0, 1024, 3, 1, 0, 52, 4)])
def test_conv_pairs_out_under_a_derived_bound(dev, case):
    """mf_conv2d_f16x2_pairs_out (round 4): the fp32 output is the plain convolution's, bit for bit; the bound is the derived one,
    bound(x1) l1_1 + bound(x2) l1_2 + max |bias|, and really bounds |y|; the pair form written by the epilogue is EXACTLY the split of the
    fp32 output under that bound (what mf_split_f16x2 gives) -- for stride-2 (BasicDown), sub-pixel (BasicUp), two-source and in-launch
    split-K plans, with per-sample operand scales 2^+-20 apart."""
    from medfusion_amd import kernels as K
    n, h, w, c1, c2, co, k, stride, ups, tile, sk = case
    scale = torch.tensor([[1.0, 2.0 ** 20, 2.0 ** -20, 3.0][i % 4] for i in range(n)]).view(n, 1, 1, 1)
    x = K.nchw_to_nhwc((_rand(f"px{case}", (n, c1, h, w)) * scale).to(dev))
    x2 = K.nchw_to_nhwc((_rand(f"py{case}", (n, c2, h, w)) * scale * 0.5).to(dev)) if c2 else None
    wt = _rand(f"pw{case}", (co, c1 + c2, k, k), 1.0 / np.sqrt((c1 + c2) * k * k)).to(dev)
    b = _rand(f"pb{case}", (co,), 0.1).to(dev)
    pad = R.monai_padding(k, stride)
    wp = K.pack_upconv_weight(wt) if ups == 2 else K.pack_conv_weight(wt)
    wh = K.split_weight_f16x2(wp)
    d = K.make_conv_desc(n, h, w, c1, c2, co, k, stride, pad, ups, tile_hint=tile, splitk_hint=sk, precision=5)
    assert K.conv_f16x2_ok(d) and K.conv_pairs_out_ok(d), case
    l1 = (float(wp[..., :c1].abs().sum(dim=(-3, -2, -1)).max()) * (1 + 1e-5), float(wp[..., c1:].abs().sum(dim=(-3, -2, -1)).max()) * (1 + 1e-5) if c2 else 0.0)
    bmax = float(b.abs().max())
    got = K.conv2d_f16x2_pairs_out(x, wh, b, d, l1, bmax, x2=x2)
    ref = K.conv2d_f16x2(x, wh, b, d, x2=x2)
    assert torch.equal(got, ref), case
    want_bound = K.bound_of(x) * l1[0] + (K.bound_of(x2) * l1[1] if c2 else 0.0) + bmax
    assert torch.allclose(got._mf_bound, want_bound, rtol=1e-6, atol=0), case
    assert bool((got._mf_bound >= ref.abs().amax(dim=(1, 2, 3))).all()), "the derived bound does not bound the output"
    slack = (got._mf_bound / ref.abs().amax(dim=(1, 2, 3))).max()
    assert float(slack) < 2.0 ** 12, float(slack)            # a few binades, far inside the 2^28 the pair format tolerates
    assert torch.equal(got._mf_split, K.split_f16x2(ref, got._mf_bound)), case
    assert torch.equal(got, K.conv2d_f16x2_pairs_out(x, wh, b, d, l1, bmax, x2=x2))   # deterministic


@pytest.mark.parametrize("case", [(16, 8, 32, 32, 256, 3), (2, 8, 64, 64, 256, 3), (3, 3, 20, 12, 64, 3), (2, 4, 8, 8, 128, 1)])
def test_input_convolution_on_the_pair_kernel(dev, case):
    """round 4: the NCHW network input goes to a zero-padded 32-channel fp16-pair operand in one launch (mf_pack_nchw_pairs_f32: the bound it
    publishes is the sample's max |x|, its pairs are EXACTLY the split of the padded NHWC tensor under that bound), and the input convolution
    (8 -> 256 at the UNet, 3 -> 64 at the VAE) runs on the fp16-pair kernel with zero-padded weights: against an fp64 convolution to the
    fp32 class, and against the fp32 direct kernel it replaces."""
    from medfusion_amd import blocks as BLK
    from medfusion_amd import kernels as K
    from medfusion_amd import lib as L
    n, c, h, w, co, k = case
    scale = torch.tensor([[1.0, 2.0 ** 12, 2.0 ** -9][i % 3] for i in range(n)]).view(n, 1, 1, 1)
    x = (_rand(f"ix{case}", (n, c, h, w)) * scale).to(dev)
    xp = K.pack_nchw_pairs(x, 32)
    assert K.pairs_only(xp) and torch.equal(xp._mf_bound, x.abs().amax(dim=(1, 2, 3)))
    ref = torch.zeros((n, h, w, 32), device=dev)
    ref[..., :c] = K.nchw_to_nhwc(x)
    assert torch.equal(xp._mf_split, K.split_f16x2(ref, xp._mf_bound))
    conv = BLK.Conv(c, co, k, 1, R.monai_padding(k, 1)).to(dev)
    S.synth_state_dict(conv, f"ic{case}.")
    want = F.conv2d(x.cpu().double(), conv.weight.detach().cpu().double(), conv.bias.detach().cpu().double(), padding=conv.pad).float()
    old = BLK.INPUT_CONV_ON_PAIRS
    try:
        BLK.INPUT_CONV_ON_PAIRS = True
        y_new = conv(x, in_layout=L.LAYOUT_NCHW)
        assert getattr(y_new, "_mf_split", None) is not None and getattr(y_new, "_mf_bound", None) is not None      # the pair form came with it
        assert torch.equal(y_new._mf_split, K.split_f16x2(y_new, y_new._mf_bound))
        BLK.INPUT_CONV_ON_PAIRS = False
        conv._descs.clear()
        y_old = conv(x, in_layout=L.LAYOUT_NCHW)
    finally:
        BLK.INPUT_CONV_ON_PAIRS = old
    e_new, e_old = relerr_rows_local(K.nhwc_to_nchw(y_new), want), relerr_rows_local(K.nhwc_to_nchw(y_old), want)
    assert e_new < 1e-6 and e_new < 3 * e_old + 2e-7, (case, e_new, e_old)


def relerr_rows_local(a, b):
    a, b = a.detach().cpu().double().reshape(a.shape[0], -1), b.detach().cpu().double().reshape(b.shape[0], -1)
    return float(((a - b).abs().amax(1) / b.abs().amax(1)).max())


FUSED_CASES = [
    # (N, H, W, C1, C2, Cout, k, G, tile, split-K, residual, emb, out_fp32)   residual: none | f32 (identity, bound known) | pairs (pairs-only) | slots (a conv_res output)
    (16, 32, 32, 256, 0, 256, 3, 32, 52, 1, "f32", True, True),       # the dominant launch of cfg2: 256 workgroups, 16 tiles per sample
    (16, 32, 32, 256, 0, 256, 3, 32, 52, 1, "pairs", False, False),   # ... as the second convolution of a ResBlock (residual and output as pairs only)
    (16, 16, 16, 512, 0, 512, 3, 32, 52, 2, "pairs", True, False),    # in-launch split-K in front of the rendezvous
    (16, 16, 16, 256, 0, 512, 3, 32, 36, 1, "slots", True, True),     # 8-wave 128 x 64 tile, residual bound from the slots of a 1x1 conv_res
    (16, 8, 8, 1024, 0, 1024, 3, 32, 52, 4, "f32", True, True),       # TWO samples per tile (HW = 64 < BM = 128), two tree levels
    (16, 8, 8, 1024, 1024, 1024, 3, 32, 51, 8, "slots", False, True), # two-source concat, 512 workgroups two per CU
    (16, 8, 8, 512, 0, 1024, 3, 32, 53, 2, "none", True, True),       # 64-row tile: one sample per tile at 8 x 8
    (16, 32, 32, 256, 256, 256, 3, 32, 62, 2, "slots", True, False),  # halo tile (256 rows), two-source
    (16, 32, 32, 512, 0, 256, 3, 32, 0, 0, "f32", False, True),       # the planner's own choice
    (2, 8, 8, 64, 32, 128, 3, 8, 0, 0, "f32", True, True),            # a tiny launch (two workgroups), G = 8
    (3, 16, 16, 64, 0, 128, 1, 8, 0, 0, "none", False, True),         # 1x1, N = 3
    (5, 8, 8, 96, 32, 256, 3, 32, 32, 1, "f32", True, True),          # 256-row tile holding FOUR 8 x 8 samples, the last tile ragged (N = 5)
]


def _fused_operands(case, dev):
    from medfusion_amd import kernels as K
    n, h, w, c1, c2, co, k, G, tile, sk, rkind, use_emb, out_fp32 = case
    x = K.nchw_to_nhwc((_rand(f"fx{case}", (n, c1, h, w)) * torch.tensor([[1.0, 300.0, 1e-3, 7.0][i % 4] for i in range(n)]).view(n, 1, 1, 1)).to(dev))
    x2 = K.nchw_to_nhwc(_rand(f"fy{case}", (n, c2, h, w)).to(dev)) if c2 else None
    wh = K.split_weight_f16x2(K.pack_conv_weight(_rand(f"fw{case}", (co, c1 + c2, k, k), 1.0 / np.sqrt((c1 + c2) * k * k)).to(dev)))
    b = _rand(f"fb{case}", (co,), 0.1).to(dev)
    gamma, beta = (1 + 0.2 * _rand(f"fg{case}", (co,))).to(dev), _rand(f"fz{case}", (co,), 0.1).to(dev)
    emb = _rand(f"fe{case}", (n, co)).to(dev) if use_emb else None
    pad = R.monai_padding(k, 1)
    d = K.make_conv_desc(n, h, w, c1, c2, co, k, 1, pad, 0, tile_hint=tile, splitk_hint=sk, precision=5)

    def residual():
        if rkind == "none":
            return None
        if rkind == "slots":   # the output of a 1x1 conv_res that measured its own bound: fp32 values + per-(tile, wave) slot maxima
            w1 = K.split_weight_f16x2(K.pack_conv_weight(_rand(f"fr{case}", (co, c1 + c2, 1, 1), 1.0 / np.sqrt(c1 + c2)).to(dev)))
            d1 = K.make_conv_desc(n, h, w, c1, c2, co, 1, 1, 0, 0, precision=5)
            r = K.conv2d_f16x2(x, w1, b, d1, x2=x2, measure_out=True)
            assert getattr(r, "_mf_slots", None) is not None
            return r
        r = K.nchw_to_nhwc((2.0 * _rand(f"fq{case}", (n, co, h, w))).to(dev))
        if rkind == "pairs":   # exists as fp16 pairs only (the output of the previous apply pass of a ResBlock)
            K.split_of(r)
            r._mf_pairs_only = True
        return r

    bc = float(gamma.abs().max()) * (h * w * co // G) ** 0.5 + float(beta.abs().max())
    return x, x2, wh, b, gamma, beta, emb, d, residual, bc


@pytest.mark.parametrize("case", FUSED_CASES, ids=lambda c: "x".join(str(v) for v in c))
def test_conv_gn_apply_in_one_launch_equals_the_two_launch_form(dev, case):
    """mf_conv2d_f16x2_gn_apply (round 4: the GroupNorm + Swish + residual + embedding pass inside the convolution's launch, the tiles of a
    sample meeting at a counter) against mf_conv2d_f16x2 + mf_gn_apply_from_partials_pairs_f32 on the same operands: the fp16-pair output,
    its bound and -- where written -- the fp32 output must be IDENTICAL (same records, same summation order, same per-element operations);
    300 further launches alternating between two inputs check that every launch leaves its counters zero and that nothing races."""
    from medfusion_amd import kernels as K
    n, h, w, c1, c2, co, k, G, tile, sk, rkind, use_emb, out_fp32 = case
    x, x2, wh, b, gamma, beta, emb, d, residual, bc = _fused_operands(case, dev)
    parts, words = K.conv_gn_parts(d, G), K.conv_fuse_words(d, G)
    if tile == 0 and (parts == 0 or words == 0):
        pytest.skip("the planner's own plan for this shape has no fused form")
    assert parts > 0 and words == 2 * n, (case, parts, words)

    def two_launches(xin):
        res = residual()
        y, partial = K.conv2d_f16x2(xin, wh, b, d, x2=x2, gn_groups=G, gn_parts=parts)
        return K.gn_apply(y, K.GnPartials(partial, parts, 1e-5), gamma, beta, G, 1, res, emb, emb.stride(0) if emb is not None else 0, out=y, split=True,
                          bconst=bc, out_fp32=out_fp32)

    def one_launch(xin):
        return K.conv2d_f16x2_gn_apply(xin, wh, b, d, gamma, beta, G, 1e-5, parts, words, act=1, residual=residual(), emb=emb,
                                       emb_stride=emb.stride(0) if emb is not None else 0, x2=x2, bconst=bc, out_fp32=out_fp32)

    def decoded(t):   # pair tensor on the host: (hi + lo / 2048) 2^s, as NCHW fp64
        return (_decode_pairs(t._mf_split, t.shape) * torch.exp2(_scale_exp(t._mf_bound)).double().view(-1, 1, 1, 1)).permute(0, 3, 1, 2)

    def want64(xin):   # GroupNorm, Swish, residual, embedding in fp64 on the convolution's fp32 output (the convolution itself: test_conv_f16x2)
        res = residual()
        resv = None if res is None else (decoded(res) if rkind == "pairs" else K.nhwc_to_nchw(res).cpu().double())
        y64 = K.nhwc_to_nchw(K.conv2d_f16x2(xin, wh, b, d, x2=x2)).cpu().double()
        yn = F.group_norm(y64, G, gamma.cpu().double(), beta.cpu().double(), 1e-5)
        want = yn * torch.sigmoid(yn)
        if resv is not None:
            want = want + resv
        if emb is not None:
            want = want + emb.cpu().double().view(n, co, 1, 1)
        return want

    def explain(got, ref, xin):   # where the two forms differ, and which of them is right
        neq = (got._mf_split != ref._mf_split).cpu()
        words = neq.reshape(-1, co // 8, 8).sum(dim=(0, 1)).tolist()                 # by 32-bit word of the 32-byte group: hi 01 23 45 67 | lo 01 23 45 67
        pix = neq.reshape(n, h * w, co).any(-1)
        rows = pix.reshape(-1)[: min(n * h * w, 256)].reshape(-1, 32).sum(1).tolist() # differing pixels per 32-row block of the first tile(s)
        chan = neq.reshape(-1, co).any(0).reshape(-1, 8).sum(1).tolist()             # differing channels per octet
        w64 = want64(xin)
        return (f"differ: {int(neq.sum())} of {neq.numel()} words; by word of the group {words}; samples with differences {pix.any(1).tolist()}; per 32-pixel block "
                f"{rows}; octets {chan[:16]}; error vs fp64: one launch {relerr(decoded(got), w64):.2e}, two launches {relerr(decoded(ref), w64):.2e}; "
                f"bounds equal {torch.equal(got._mf_bound, ref._mf_bound)}")

    xb = x * 1.5 + 0.25
    K.split_of(xb)
    refs = [two_launches(x), two_launches(xb)]
    for it in range(302):
        which = it & 1
        got = one_launch(xb if which else x)
        if it < 2 or it % 50 == 0 or it == 301:
            ref = refs[which]
            assert K.pairs_only(got) == (not out_fp32)
            if not (torch.equal(got._mf_split, ref._mf_split) and torch.equal(got._mf_bound, ref._mf_bound)):
                pytest.fail(f"{case} launch {it}: " + explain(got, ref, xb if which else x))
            if out_fp32:
                assert torch.equal(got, ref), (case, it)
    rv = K.Rendezvous.get(words, dev)
    torch.cuda.synchronize()
    assert int(rv.abs().sum().item()) == 0, "a launch left its rendezvous counters (or the error flag) non-zero"
    assert relerr(decoded(one_launch(x)), want64(x)) < 2e-6, case


def _scale_exp(bound):
    """the exponent s of the per-sample scale 2^s (csrc/split_f16.h: floor(log2 bound) - 14)"""
    return (torch.floor(torch.log2(bound.double())) - 14).clamp(-100, 100).float().cpu()


def test_conv_gn_apply_fused_refuses_what_it_cannot_do(dev):
    """plans whose workgroups are not all resident at once (or that reduce split-K through a reducer pass) have no fused form:
    mf_conv2d_f16x2_fuse_words says 0 and the entry point refuses"""
    from medfusion_amd import kernels as K
    big = K.make_conv_desc(16, 128, 128, 128, 0, 128, 3, 1, 1, 0, precision=5)          # VAE 128^2 level: thousands of workgroups
    assert K.conv_f16x2_ok(big) and K.conv_fuse_words(big, 32) == 0
    odd = K.make_conv_desc(16, 16, 16, 384, 0, 512, 3, 1, 1, 0, tile_hint=52, splitk_hint=3, precision=5)   # split-K 3: slabs + reducer pass
    assert K.conv_f16x2_ok(odd) and K.conv_fuse_words(odd, 32) == 0
    ok = K.make_conv_desc(16, 32, 32, 256, 0, 256, 3, 1, 1, 0, precision=5)
    assert K.conv_fuse_words(ok, 32) == 32


FAULT = r"""
import os, sys, warnings, torch
sys.path.insert(0, {root!r})
import medfusion_amd as M
from medfusion_amd import kernels as K, published as P
from medfusion_amd.unet import UNet, TimeEmbbeding
dev = torch.device("cuda:0")
ukw = dict(in_ch=8, out_ch=8, spatial_dims=2, hid_chs=[256, 256, 256, 512], kernel_sizes=[3] * 4, strides=[1, 2, 2, 2], time_embedder=TimeEmbbeding,
           time_embedder_kwargs={{"emb_dim": 64}}, cond_embedder=None, deep_supervision=False, use_res_block=True, use_attention="none")
pipe = M.DiffusionPipeline(M.GaussianNoiseScheduler, UNet, None, P.published_scheduler_kwargs(), ukw, estimator_objective="x_T", clip_x0=False)
P.seeded_fill(pipe.noise_estimator, "fault.unet.")   # (widths >= 256: 32 groups of >= 8 channels, what the convolution's epilogue can take statistics of)
pipe.to(dev).eval()
with warnings.catch_warnings(record=True) as wl:
    warnings.simplefilter("always")
    a = pipe.sample(2, (8, 16, 16), noise=M.PhiloxDeviceNoise(9), steps=6, use_ddim=True)
torch.cuda.synchronize()
print("FAULT_MODE", os.environ.get("MEDFUSION_FUSE_FAULT"), "disabled after:", K.Rendezvous.disabled, "warnings:", len([w for w in wl if "rendezvous" in str(w.message)]),
      "fused launches:", K.Rendezvous.launches)
torch.save(a.cpu(), sys.argv[1])
"""


def test_fused_rendezvous_timeout_falls_back_to_the_two_launch_form(dev, tmp_path):
    """MEDFUSION_FUSE_FAULT=1 makes every rendezvous wait for one arrival too many: the first fused launch times out (50 ms), raises the error
    flag, every later one stops waiting; sample() sees the flag at the end of its loop, switches the fused form off, rewinds the noise
    counter and re-runs -- the images equal those of a process that never fused (the default) and of a healthy fused run (the opt-in
    MEDFUSION_FUSED_APPLY=1)."""
    import subprocess
    script = tmp_path / "fault.py"
    script.write_text(FAULT.format(root=str(ROOT)))
    outs = {}
    for tag, env_extra in (("fault", {"MEDFUSION_FUSE_FAULT": "1", "MEDFUSION_FUSED_APPLY": "1"}), ("off", {"MEDFUSION_FUSED_APPLY": "0"}), ("on", {"MEDFUSION_FUSED_APPLY": "1"})):
        env = dict(os.environ, MEDFUSION_WINOGRAD="0", **env_extra)    # (a feature of the direct form: the Winograd form has its own tail)
        r = subprocess.run([sys.executable, str(script), str(tmp_path / f"{tag}.pt")], env=env, capture_output=True, text=True, timeout=900, cwd=str(ROOT))
        assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
        outs[tag] = (torch.load(tmp_path / f"{tag}.pt"), [ln for ln in r.stdout.splitlines() if ln.startswith("FAULT_MODE")][-1])
    assert "disabled after: True warnings: 1" in outs["fault"][1], outs["fault"][1]
    assert "disabled after: False" in outs["on"][1] and "disabled after: True" in outs["off"][1]
    assert int(outs["on"][1].rsplit(":", 1)[1]) > 0 and int(outs["off"][1].rsplit(":", 1)[1]) == 0   # the healthy run did fuse, the switched-off one never
    assert torch.equal(outs["fault"][0], outs["off"][0])
    assert torch.equal(outs["on"][0], outs["off"][0])       # the fused form IS the two-launch form, bit for bit, through a whole sampling loop


@pytest.mark.parametrize("shape", [(16, 32, 32, 256, 256, 256), (16, 16, 16, 512, 512, 512), (16, 8, 8, 1024, 1024, 1024), (16, 8, 8, 1024, 512, 512)])
def test_conv_f16x2_two_source_1x1_at_published_sizes(dev, shape):
    """conv_res of the out-blocks at the cfg2 batch (two-source concat, planner's own tile / split-K incl. the in-launch reduction): exact to
    the fp32 class against an fp64 product computed on the device, and the same bits on every launch.  (Regression: one build of the
    epilogue produced the bare bias in 16 lanes of one output channel, now and then, on the first of these shapes.)"""
    from medfusion_amd import kernels as K
    n, h, w, c1, c2, co = shape
    g = torch.Generator().manual_seed(11)
    x, x2 = torch.randn((n, h, w, c1), generator=g).to(dev), torch.randn((n, h, w, c2), generator=g).to(dev)
    wt = (torch.randn((co, 1, 1, c1 + c2), generator=g) * 0.02).to(dev)
    b = torch.randn((co,), generator=g).to(dev)
    want = (torch.cat([x, x2], -1).double().reshape(-1, c1 + c2) @ wt.double().reshape(co, -1).T + b.double()).reshape(n, h, w, co)
    wh = K.split_weight_f16x2(wt)
    d = K.make_conv_desc(n, h, w, c1, c2, co, 1, 1, 0, 0, precision=5)
    first = None
    for rep in range(12):
        y = K.conv2d_f16x2(x, wh, b, d, x2=x2, measure_out=bool(rep % 2))
        assert float((y.double() - want).abs().max() / want.abs().max()) < 2e-6, (shape, rep)
        if first is None:
            first = y.clone()
        assert torch.equal(y, first), (shape, rep)


@pytest.mark.parametrize("case", [(16, 16, 16, 512, 0, 512, 3, 53, 2), (16, 16, 16, 512, 0, 512, 3, 54, 2), (16, 16, 16, 512, 0, 512, 3, 36, 2),
                                  (16, 16, 16, 512, 0, 512, 3, 33, 2), (16, 16, 16, 1024, 0, 512, 3, 51, 4), (16, 8, 8, 1536, 0, 512, 3, 53, 8),
                                  (16, 8, 8, 1024, 1024, 1024, 3, 31, 8), (16, 32, 32, 256, 256, 256, 3, 32, 2), (16, 16, 16, 512, 512, 512, 1, 53, 2)])
def test_conv_f16x2_split_k_meets_inside_the_launch(dev, case):
    """the in-launch split-K reduction under the condition that exposes a stale or lost hand-off: launches ALTERNATE between different
    inputs (a tile read too early, through a stale cache line, or not added at all shows up against the un-split launch of the same input),
    and equal inputs give equal bits.  Shapes and (tile, split) pairs of the cfg2 batch."""
    from medfusion_amd import kernels as K
    n, h, w, c1, c2, co, k, tile, sk = case
    g = torch.Generator().manual_seed(5)
    pad = 1 if k == 3 else 0
    wt = (torch.randn((co, k, k, c1 + c2), generator=g) * 0.02).to(dev)
    b = torch.randn((co,), generator=g).to(dev)
    wh = K.split_weight_f16x2(wt)
    d = K.make_conv_desc(n, h, w, c1, c2, co, k, 1, pad, 0, tile_hint=tile, splitk_hint=sk, precision=5)
    d1 = K.make_conv_desc(n, h, w, c1, c2, co, k, 1, pad, 0, tile_hint=tile, splitk_hint=1, precision=5)
    xs = [(torch.randn((n, h, w, c1), generator=g).to(dev) * (1 + 3 * i), torch.randn((n, h, w, c2), generator=g).to(dev) if c2 else None) for i in range(2)]
    refs = [K.conv2d_f16x2(x, wh, b, d1, x2=x2).clone() for x, x2 in xs]
    firsts = [None, None]
    for rep in range(16):
        i = (rep + rep // 3) % 2
        y = K.conv2d_f16x2(xs[i][0], wh, b, d, x2=xs[i][1], measure_out=bool(rep % 2))
        assert float((y - refs[i]).abs().max() / refs[i].abs().max()) < 1e-5, (case, rep)
        if firsts[i] is None:
            firsts[i] = y.clone()
        assert torch.equal(y, firsts[i]), (case, rep)


def _stress_cases():
    import importlib.util
    spec = importlib.util.spec_from_file_location("conv_stress", str(ROOT / "scripts" / "conv_stress.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


@pytest.mark.parametrize("case", _stress_cases().CASES, ids=lambda c: "x".join(str(v) for v in c))
def test_conv_f16x2_stress(dev, case):
    """2000 launches per (shape, tile, split-K) of the cfg2 batch, alternating between three inputs, every one compared on the device with an
    un-split launch on another tile (bit for bit where the plan does not split, else to 1e-5 and bit for bit with the first result of the
    same input): the test that shows the lost split-K partials of the packed-fp32 build in every launch (profiles/r03_pk_repro.txt) and
    must stay at zero on the product build.  Includes the two-workgroups-per-CU tiles with and without split-K."""
    bad, plan, desc = _stress_cases().run_case(case, 2000, dev)
    assert bad == 0, (case, plan, bad, desc)


def _stress_subprocess(env_extra, args):
    import subprocess
    env = dict(os.environ, **env_extra)
    r = subprocess.run([sys.executable, str(ROOT / "scripts" / "conv_stress.py"), *args], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    tot = [ln for ln in r.stdout.splitlines() if ln.startswith("TOTAL bad launches:")]
    assert tot, r.stdout[-2000:]
    return int(tot[0].split(":")[1]), r.stdout


def test_conv_f16x2_stress_with_fences_around_the_pair_counter(dev):
    """MF_CONV_TREE=2: the same hand-off with the memory model's release / acquire fences around the counter (conv_f16x2.h) -- same bits,
    zero bad launches (the default form relies on sc1 stores + sc1 loads; profiles/r03_tree_fence_cost.txt has what the fences cost)"""
    bad, out = _stress_subprocess({"MF_CONV_TREE": "2"}, ["--reps", "600", "--tiles", "53,54,36,31,51"])
    assert bad == 0, out[-3000:]


def test_packed_fp32_twin_of_the_convolution(dev):
    """The same stress on the twin of the library whose conv_f16x2.hip is compiled WITH packed fp32 (what round 2 switched off).  Its device
    code contains the operand selection of the gfx950 erratum (the ISA lint flags it); wherever that twin loses split-K partials the lint
    must have flagged it -- and the product build, which the lint passes, is clean in the test above."""
    from medfusion_amd import build as B
    import re
    twin = B.build_variant("pk", packed_fp32=True)
    cc = B.hipcc()
    s_path = B.OBJ / "variants" / "conv_f16x2_pk.s"
    B._run([cc, *B.CFLAGS, "--cuda-device-only", "-S", str(B.CSRC / "conv_f16x2.hip"), "-o", str(s_path)], False)
    flagged = sum(1 for ln in s_path.read_text().splitlines() if B._PK_SRC1_HIGH.match(ln))
    bad, out = _stress_subprocess({"MEDFUSION_LIB": str(twin)}, ["--reps", "300", "--tiles", "53,54,36"])
    print(f"packed-fp32 twin: {flagged} flagged instructions, {bad} bad launches of the stress")
    assert bad == 0 or flagged > 0, out[-3000:]


from tests.util import GROUP_CASES  # noqa: E402


@pytest.mark.parametrize("case", GROUP_CASES)
def test_two_convolutions_in_one_launch_equal_the_two_launches(dev, case):
    """mf_conv2d_f16x2_group (round 4): the 3x3 of a channel-changing ResBlock with GroupNorm records and its conv_res 1x1 with measured bound
    slots, both on the same (two-source) input, in ONE launch -- outputs, records and slots bit-identical to the two mf_conv2d_f16x2 launches,
    on every host / guest tile pair that is instantiated (plain + plain, halo + plain, in-launch split-K on either side, an odd chunk count),
    over repeated launches (the split-K counters come back to zero)."""
    from medfusion_amd import kernels as K
    n, h, w, c1, c2, co, ta, ska, tb = case
    G = 32 if co >= 256 else 8
    scale = torch.tensor([[1.0, 2.0 ** 9, 2.0 ** -7][i % 3] for i in range(n)]).view(n, 1, 1, 1)
    x1 = K.nchw_to_nhwc((_rand(f"gx{case}", (n, c1, h, w)) * scale).to(dev))
    x2 = K.nchw_to_nhwc((_rand(f"gy{case}", (n, c2, h, w)) * scale * 0.5).to(dev)) if c2 else None
    w3 = _rand(f"gw3{case}", (co, c1 + c2, 3, 3), 1.0 / np.sqrt((c1 + c2) * 9)).to(dev)
    w1 = _rand(f"gw1{case}", (co, c1 + c2, 1, 1), 1.0 / np.sqrt(c1 + c2)).to(dev)
    b3, b1 = _rand(f"gb3{case}", (co,), 0.1).to(dev), _rand(f"gb1{case}", (co,), 0.1).to(dev)
    wh3, wh1 = K.split_weight_f16x2(K.pack_conv_weight(w3)), K.split_weight_f16x2(K.pack_conv_weight(w1))
    da = K.make_conv_desc(n, h, w, c1, c2, co, 3, 1, 1, 0, tile_hint=ta, splitk_hint=ska, precision=5)
    db = K.make_conv_desc(n, h, w, c1, c2, co, 1, 1, 0, 0, tile_hint=tb, precision=5)
    assert K.conv_f16x2_ok(da) and K.conv_f16x2_ok(db), case
    pa, pb = K.pin_conv_plan(da), K.pin_conv_plan(db)
    parts = K.conv_gn_parts(da, G)
    assert parts > 0 and pb[1] > 0, case
    assert K.conv_group_ok(da, G, db, 0), (case, K.conv_plan(da), K.conv_plan(db))
    y_ref, part_ref = K.conv2d_f16x2(x1, wh3, b3, da, x2=x2, gn_groups=G, gn_parts=parts, pinned=pa)
    r_ref = K.conv2d_f16x2(x1, wh1, b1, db, x2=x2, measure_out=True, pinned=pb)
    for rep in range(3):
        (y, part), r = K.conv2d_f16x2_group(x1, x2, dict(w_split=wh3, bias=b3, d=da, gn_groups=G, gn_parts=parts, pinned=pa), dict(w_split=wh1, bias=b1, d=db, pinned=pb))
        assert torch.equal(y, y_ref), (case, rep, "3x3 output")
        assert torch.equal(part, part_ref), (case, rep, "GroupNorm records")
        assert torch.equal(r, r_ref), (case, rep, "1x1 output")
        assert torch.equal(r._mf_slots, r_ref._mf_slots), (case, rep, "bound slots")
    assert int(K.SyncWords.get(1, x1.device).abs().max()) == 0          # every split-K counter back at zero


def test_grouped_conv_res_inside_the_blocks(dev):
    """MEDFUSION_GROUPED_CONV_RES=1 (blocks.GROUPED_CONV_RES): a channel-changing BasicResBlock gives the same bits with conv_res inside the
    3x3's launch, and falls back to two launches where the pair cannot share one (an odd shape)."""
    from medfusion_amd import blocks as BLK
    from medfusion_amd import kernels as K
    blk = BLK.BasicResBlock(2, 256, 512, 3, 1, ("GROUP", {"num_groups": 32, "affine": True}), ("Swish", {})).to(dev)
    S.synth_state_dict(blk, "grp.")
    x = K.nchw_to_nhwc(_rand("grpx", (16, 256, 16, 16)).to(dev))
    emb = _rand("grpe", (16, 512)).to(dev)
    old, old_w = BLK.GROUPED_CONV_RES, BLK.WINOGRAD
    try:
        BLK.WINOGRAD = 0        # (the direct form's grouped launch; the Winograd form's: tests/test_winograd_gpu.py)
        BLK.GROUPED_CONV_RES = False
        want = blk(x, emb=emb, emb_stride=512)
        BLK.GROUPED_CONV_RES = True
        got = blk(x, emb=emb, emb_stride=512)
        assert blk._group and next(iter(blk._group.values())) is not None      # the pair shares a launch for this shape
        assert torch.equal(got, want) and torch.equal(got._mf_split, want._mf_split) and torch.equal(got._mf_bound, want._mf_bound)
        xo = K.nchw_to_nhwc(_rand("grpo", (2, 256, 10, 12)).to(dev))              # no instantiated pair for what the planner picks here, or it is: either way equal
        BLK.GROUPED_CONV_RES = False
        want_o = blk(xo, emb=emb[:2], emb_stride=512)
        BLK.GROUPED_CONV_RES = True
        assert torch.equal(blk(xo, emb=emb[:2], emb_stride=512), want_o)
    finally:
        BLK.GROUPED_CONV_RES, BLK.WINOGRAD = old, old_w
