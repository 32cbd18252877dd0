"""The Winograd F(2x2, 3x3) form of the 3x3 stride-1 convolutions (mf_wino_*; csrc/winograd.h) on a real MI355X: every piece against an fp64
reference of the same op, the whole against an fp64 convolution AND against the direct form on the same arithmetic."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from oracle import synth as S
from tests.util import relerr


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "GPU tests need a ROCm device"
    import medfusion_amd  # noqa: F401
    return torch.device("cuda:0")


def _rand(name, shape, scale=1.0):
    return S.synth_input(name, shape, scale)


BT = torch.tensor([[1, 0, -1, 0], [0, 1, 1, 0], [0, -1, 1, 0], [0, 1, 0, -1]], dtype=torch.float64)
G = torch.tensor([[1, 0, 0], [.5, .5, .5], [.5, -.5, .5], [0, 0, 1]], dtype=torch.float64)


def _decode_pairs(xs, shape):
    """fp16-pair tensor (int32 storage, [hi x 8][lo' x 8] per 8 elements) -> fp64 values (un-scaled)"""
    raw = xs.cpu().contiguous().view(torch.float16).double().reshape(-1, 2, 8)
    return (raw[:, 0] + raw[:, 1] / 2048.0).reshape(shape)


# (N, H, W, C1, C2, Cout)
WINO_CASES = [
    (4, 8, 8, 32, 0, 128),        # N T = 64: only the 64-row tiles hold a component
    (16, 8, 8, 64, 32, 128),      # two-source (skip concat)
    (2, 16, 16, 256, 0, 256),
    (2, 16, 8, 96, 0, 512),       # non-square, Cin = 96
    (1, 32, 32, 64, 0, 64),       # Cout = 64
    (16, 8, 8, 1024, 0, 1024),    # published 8 x 8 level
    (16, 8, 8, 1024, 1024, 1024), # published out-block, K = 2048: split-K inside the launch
    (16, 16, 16, 512, 0, 512),    # published 16 x 16 level
]
TILES = {31: (128, 256), 32: (256, 128), 33: (128, 128), 34: (128, 128), 35: (256, 64), 36: (128, 64), 37: (64, 256), 51: (128, 128), 52: (128, 128),
         53: (64, 128), 54: (128, 64)}


def test_wino_weight_and_input_transforms(dev):
    from medfusion_amd import kernels as K
    co, ci = 64, 32
    w = _rand("wino_w", (co, ci, 3, 3), 0.1)
    u = K.wino_pack_weight(w.to(dev)).cpu().double().reshape(16, co, ci)
    want = torch.einsum("ia,ocab,jb->ijoc", G, w.double(), G).reshape(16, co, ci)
    assert torch.equal(u.float(), want.float())                       # fp64 arithmetic, ONE rounding to fp32
    n, h, wd, c = 3, 8, 12, 32
    x = _rand("wino_x", (n, h, wd, c)) * torch.tensor([1.0, 2.0 ** 30, 2.0 ** -30]).view(-1, 1, 1, 1)
    xd = x.to(dev)
    v, vb = K.wino_input(xd)
    assert torch.equal(vb.cpu().view(16, n), (4.0 * K.bound_of(xd).cpu()).expand(16, n))
    xq = _decode_pairs(K.split_of(xd), (n, h, wd, c))                 # the values the kernel starts from (23-bit operands, scaled per sample)
    s_x = torch.floor(torch.log2(K.bound_of(xd).cpu().double())) - 14
    xq = xq * (2.0 ** s_x).view(-1, 1, 1, 1)
    xp = F.pad(xq.permute(0, 3, 1, 2), (1, 1, 1, 1))                  # [n, c, h + 2, w + 2]
    patches = xp.unfold(2, 4, 2).unfold(3, 4, 2)                       # [n, c, h/2, w/2, 4, 4]
    V = torch.einsum("ia,nctuab,jb->ijntuc", BT, patches, BT).reshape(16, n, (h // 2) * (wd // 2), c)
    s_v = torch.floor(torch.log2(vb.cpu().double())) - 14
    got = _decode_pairs(v, (16, n, (h // 2) * (wd // 2), c)) * (2.0 ** s_v).view(16, n, 1, 1)
    err = (got - V).abs().amax(dim=(0, 2, 3)) / V.abs().amax(dim=(0, 2, 3))
    assert float(err.max()) < 2.0 ** -21, err                         # two fp32 roundings of the +-1 sums + the 23-bit split, relative to the sample's max


def _tiles_for(case):
    n, h, w, c1, c2, co = case
    rows = n * (h // 2) * (w // 2)
    return [0] + [t for t, (bm, bn) in TILES.items() if rows % bm == 0 and co % bn == 0]


@pytest.mark.parametrize("case", WINO_CASES, ids=lambda c: "x".join(str(v) for v in c))
def test_conv_wino(dev, case):
    """error against an fp64 convolution within the fp32 tolerance and of the class of the direct form on the same arithmetic; GroupNorm records
    equal to the statistics of the output; exact power-of-two scale invariance; on every tile that holds a component and split-K 1 / 2 / 4"""
    from medfusion_amd import kernels as K
    n, h, w, c1, c2, co = case
    cin = c1 + c2
    x = _rand(f"wx{case}", (n, c1, h, w))
    x2 = _rand(f"wy{case}", (n, c2, h, w)) if c2 else None
    wt = _rand(f"ww{case}", (co, cin, 3, 3), 1.0 / np.sqrt(cin * 9))
    b = _rand(f"wb{case}", (co,), 0.1)
    xin = x if x2 is None else torch.cat([x, x2], 1)
    want = F.conv2d(xin.double(), wt.double(), b.double(), padding=1).float()
    xd = K.nchw_to_nhwc(x.to(dev))
    x2d = K.nchw_to_nhwc(x2.to(dev)) if c2 else None
    wh = K.split_weight_f16x2(K.pack_conv_weight(wt.to(dev)))
    uh = K.split_weight_f16x2(K.wino_pack_weight(wt.to(dev)))
    d0 = K.make_conv_desc(n, h, w, c1, c2, co, 3, 1, 1, 0, precision=5)
    e_direct = relerr(K.nhwc_to_nchw(K.conv2d_f16x2(xd, wh, b.to(dev), d0, x2=x2d)), want)
    assert K.wino_ok(d0), case
    worst = 0.0
    for tile in _tiles_for(case):
        for sk in ([0] if tile == 0 else [1, 2, 4]):
            if sk > cin // 32:
                continue
            d = K.make_conv_desc(n, h, w, c1, c2, co, 3, 1, 1, 0, tile_hint=tile, splitk_hint=sk, precision=5)
            assert K.wino_ok(d), (case, tile, sk)
            y = K.conv2d_wino_f16x2(xd, uh, b.to(dev), d, x2=x2d)
            e = relerr(K.nhwc_to_nchw(y), want)
            worst = max(worst, e)
            assert e < 1e-5 and e < 6 * e_direct + 1e-6, (case, tile, sk, e, e_direct)
            assert torch.equal(y, K.conv2d_wino_f16x2(xd, uh, b.to(dev), d, x2=x2d)), (case, tile, sk)   # bit-reproducible (in-launch split-K included)
            Gn = 32 if co % 128 == 0 else 8
            parts = K.wino_gn_parts(d, Gn)
            assert parts > 0, (case, Gn)
            y2, partial = K.conv2d_wino_f16x2(xd, uh, b.to(dev), d, x2=x2d, gn_groups=Gn, gn_parts=parts)
            assert torch.equal(y2, y)
            yg = y.double().cpu().reshape(n, -1, Gn, co // Gn)
            cnt = yg.shape[1] * yg.shape[3]
            mean, msq = yg.sum(dim=(1, 3)) / cnt, (yg * yg).sum(dim=(1, 3)) / cnt
            got = partial.sum(1).cpu() / cnt
            assert torch.allclose(got[..., 1], msq, rtol=1e-5, atol=0), (case, tile, sk)
            assert torch.allclose(got[..., 0], mean, rtol=0, atol=1e-5 * float(msq.max().sqrt())), (case, tile, sk)
    print(f"[measured] winograd {case}: max-norm rel err vs fp64 {worst:.2e} (direct form on the same arithmetic: {e_direct:.2e})")
    # operands of any magnitude: inputs scaled by 2^40 / 2^-40 per sample give exactly the scaled result (every transform is a +-1 sum)
    d = K.make_conv_desc(n, h, w, c1, c2, co, 3, 1, 1, 0, precision=5)
    y = K.conv2d_wino_f16x2(xd, uh, None, d, x2=x2d)
    sc = torch.tensor([2.0 ** 40 if i % 2 == 0 else 2.0 ** -40 for i in range(n)], device=dev).view(-1, 1, 1, 1)
    ys = K.conv2d_wino_f16x2(xd * sc, uh, None, d, x2=None if x2d is None else x2d * sc)
    assert torch.equal(ys, y * sc), case
    if x2d is not None:   # the two sources of a fused concat carry their own scales
        yb = K.nhwc_to_nchw(K.conv2d_wino_f16x2(xd * 1024.0, uh, b.to(dev), d, x2=x2d * (1.0 / 4096.0)))
        wantb = F.conv2d(torch.cat([x * 1024.0, x2 / 4096.0], 1).double(), wt.double(), b.double(), padding=1).float()
        assert relerr(yb, wantb) < 1e-5, case


def test_wino_refuses_what_it_cannot_do(dev):
    from medfusion_amd import kernels as K
    mk = lambda **kw: K.make_conv_desc(**{**dict(N=4, Hin=8, Win=8, C1=64, C2=0, Cout=128, k=3, stride=1, pad=1, upsample=0, precision=5), **kw})
    assert K.wino_ok(mk())
    for bad in (dict(k=1, pad=0), dict(stride=2), dict(Hin=7), dict(C1=48), dict(Cout=192), dict(upsample=2), dict(precision=0), dict(precision=6), dict(N=1)):
        assert not K.wino_ok(mk(**bad)), bad
    assert K.wino_gn_parts(mk(), 32) > 0 and K.wino_gn_parts(mk(), 64) == 0     # (groups of 2 channels: not a float4)
    x = K.nchw_to_nhwc(_rand("wr", (1, 64, 8, 8)).to(dev))
    uh = K.split_weight_f16x2(K.wino_pack_weight(_rand("wrw", (128, 64, 3, 3), 0.1).to(dev)))
    with pytest.raises(RuntimeError, match="Winograd path"):
        K.conv2d_wino_f16x2(x, uh, None, mk(N=1))


@pytest.mark.parametrize("shape", [(16, 8, 8, 512, 0, 1024), (16, 8, 8, 1024, 512, 512), (4, 16, 16, 64, 0, 128)])
def test_resblock_on_the_winograd_form(dev, shape):
    """a whole UnetResBlock (conv -> GroupNorm -> Swish -> + conv_res -> + emb -> conv -> ...) with its 3x3 convolutions on the Winograd form
    against the same block on the direct form: same values to fp32 rounding"""
    from medfusion_amd import blocks as BLK, kernels as K
    n, h, w, c1, c2, co = shape
    blk = BLK.UnetResBlock(2, c1 + c2, co, 3, 1, ("GROUP", {"num_groups": 32, "affine": True}), ("Swish", {}), None, emb_channels=64).to(dev)
    S.synth_state_dict(blk, f"winoblk{shape}.")
    x1 = K.nchw_to_nhwc(_rand(f"wbx{shape}", (n, c1, h, w)).to(dev))
    x2 = K.nchw_to_nhwc(_rand(f"wby{shape}", (n, c2, h, w)).to(dev)) if c2 else None
    emb = blk.local_embed(_rand(f"wbe{shape}", (n, 64)).to(dev))
    outs = {}
    old = BLK.WINOGRAD
    try:
        for mode in (0, 2):
            BLK.WINOGRAD = mode
            outs[mode] = blk(x1 if x2 is None else (x1, x2), emb).clone()
    finally:
        BLK.WINOGRAD = old
    e = relerr(outs[2], outs[0])
    print(f"[measured] UnetResBlock {shape}: Winograd vs direct form, max-norm rel diff {e:.2e}")
    assert 0 < e < 2e-5, e     # (0 would mean the Winograd form did not run)


TAIL_CASES = [
    # (N, H, W, C1, C2, Cout, G, residual kind, emb, affine, act, out_fp32)
    (16, 8, 8, 1024, 0, 1024, 32, "f32", True, True, 1, True),
    (16, 8, 8, 1024, 512, 512, 32, "slots", True, True, 1, False),
    (16, 16, 16, 512, 0, 512, 32, "pairs", False, True, 1, False),
    (4, 16, 16, 64, 0, 128, 8, None, False, False, 0, True),
    (2, 32, 32, 64, 0, 256, 32, "f32", True, True, 1, True),
    (8, 8, 12, 96, 0, 128, 8, "pairs", True, True, 1, True),
    (2, 32, 32, 64, 0, 512, 32, "pairs", True, True, 1, False),    # exactly 64 KB of dynamic LDS (1024 pixels x 16 channels): the published 512-channel tail at latent 64
    (69, 16, 16, 128, 0, 256, 32, "pairs", False, True, 1, False),   # an odd batch: the tail chunk of the reference's bulk generator (7869 % 200)
]


@pytest.mark.parametrize("case", TAIL_CASES, ids=lambda c: "x".join(str(v) for v in c))
def test_conv_wino_gn_apply_tail(dev, case):
    """the Winograd convolution with GroupNorm + Swish + residual + embedding (+ the next input transform) in the launch behind the GEMM
    (mf_conv2d_wino_gn_apply_f16x2) against (a) the three-launch Winograd form followed by the apply pass -- same values up to the rounding of
    mean / rstd (the statistics are summed in another order) -- and (b) an fp64 evaluation of the whole chain; its transform-domain output equals
    the stand-alone input transform of its own fp16-pair output bit for bit."""
    import ctypes as C
    from medfusion_amd import kernels as K, lib as L
    n, h, w, c1, c2, co, G, rkind, has_emb, affine, act, out_fp32 = case
    cin = c1 + c2
    x = _rand(f"tx{case}", (n, c1, h, w))
    x2 = _rand(f"ty{case}", (n, c2, h, w)) if c2 else None
    wt = _rand(f"tw{case}", (co, cin, 3, 3), 1.0 / np.sqrt(cin * 9))
    b = _rand(f"tb{case}", (co,), 0.1)
    gamma = (1.0 + 0.3 * _rand(f"tg{case}", (co,))) if affine else None
    beta = 0.2 * _rand(f"tbe{case}", (co,)) if affine else None
    res = _rand(f"tr{case}", (n, co, h, w), 2.0) if rkind else None
    emb = _rand(f"te{case}", (n, co), 0.5) if has_emb else None
    xd = K.nchw_to_nhwc(x.to(dev))
    x2d = K.nchw_to_nhwc(x2.to(dev)) if c2 else None
    uh = K.split_weight_f16x2(K.wino_pack_weight(wt.to(dev)))
    d = K.make_conv_desc(n, h, w, c1, c2, co, 3, 1, 1, 0, precision=5)
    assert K.wino_tail_ok(d, G), case
    gd, bd = (gamma.to(dev), beta.to(dev)) if affine else (None, None)
    embd = emb.to(dev) if has_emb else None
    cpg = co // G
    bconst = (float(gamma.abs().max()) * (h * w * cpg) ** 0.5 + float(beta.abs().max())) if affine else float((h * w * cpg) ** 0.5)

    def residual():
        if rkind is None:
            return None
        r = K.nchw_to_nhwc(res.to(dev))
        if rkind == "pairs":      # a tensor that exists as fp16 pairs only
            K.split_of(r)
            r._mf_pairs_only = True
        elif rkind == "slots":    # a measured convolution output: bound still as slot maxima
            sl = torch.zeros((n, 7), device=dev)
            sl[:, 3] = r.abs().amax(dim=(1, 2, 3))
            r._mf_slots = sl
            K._stamp(r)
        return r
    # (a) three launches + the apply pass
    parts = K.wino_gn_parts(d, G)
    y, partial = K.conv2d_wino_f16x2(xd, uh, b.to(dev), d, x2=x2d, gn_groups=G, gn_parts=parts)
    ref = K.gn_apply(y, K.GnPartials(partial, parts, 1e-5), gd, bd, G, act, residual(), embd, embd.stride(0) if has_emb else 0, split=True, bconst=bconst)
    got = K.conv2d_wino_gn_apply(xd, uh, b.to(dev), d, gd, bd, G, 1e-5, act=act, residual=residual(), emb=embd, emb_stride=embd.stride(0) if has_emb else 0,
                                 x2=x2d, bconst=bconst, out_fp32=out_fp32, want_wino=True)
    assert torch.equal(got._mf_bound, ref._mf_bound)
    s_o = torch.floor(torch.log2(got._mf_bound.cpu().double())) - 14
    gp = _decode_pairs(got._mf_split, (n, h, w, co)) * (2.0 ** s_o).view(-1, 1, 1, 1)
    rp = _decode_pairs(ref._mf_split, (n, h, w, co)) * (2.0 ** s_o).view(-1, 1, 1, 1)
    e_pairs = float((gp - rp).abs().max() / rp.abs().max())
    assert e_pairs < 2e-6, (case, e_pairs)
    if out_fp32:
        assert relerr(got, ref) < 2e-6, case
        assert float((got.cpu().double() - gp).abs().max() / rp.abs().max()) < 2.0 ** -22      # the pair mirror is the split of the fp32 output
    else:
        assert K.pairs_only(got)
    # (b) the chain in fp64
    xin = x if x2 is None else torch.cat([x, x2], 1)
    y64 = F.conv2d(xin.double(), wt.double(), b.double(), padding=1)
    t64 = F.group_norm(y64, G, gamma.double() if affine else None, beta.double() if affine else None, 1e-5)
    if act:
        t64 = t64 * torch.sigmoid(t64)
    if rkind:
        t64 = t64 + res.double()
    if has_emb:
        t64 = t64 + emb.double()[:, :, None, None]
    e64 = float((gp.permute(0, 3, 1, 2) - t64).abs().max() / t64.abs().max())
    print(f"[measured] winograd conv + GroupNorm tail {case}: vs fp64 chain {e64:.2e}; vs the three-launch form {e_pairs:.2e}")
    assert e64 < 1e-5, (case, e64)
    # the transform-domain output: what the stand-alone transform makes of the pair output, bit for bit
    T = (h // 2) * (w // 2)
    v = torch.empty((16, n, T, co), dtype=torch.int32, device=dev)
    vb = torch.empty((16 * n,), dtype=torch.float32, device=dev)
    L.check(L.load().mf_wino_input_f16x2(got._mf_split.data_ptr(), got._mf_bound.data_ptr(), v.data_ptr(), vb.data_ptr(), n, h, w, co, K.stream()), "wino_input")
    assert torch.equal(vb, got._mf_wino_bound) and torch.equal(v, got._mf_wino), case


def test_winograd_sites_learn_to_write_the_transform_domain(dev):
    """a ResBlock chain on the Winograd form: the first evaluation transforms the tail outputs with the stand-alone pass and marks the producing
    sites; the second has them written by the tails -- same bits, fewer launches"""
    import ctypes as C
    from medfusion_amd import blocks as BLK, kernels as K, lib as L
    n, h, c = 16, 8, 512
    blk = BLK.UnetResBlock(2, c, c, 3, 1, ("GROUP", {"num_groups": 32, "affine": True}), ("Swish", {}), None, emb_channels=64).to(dev)
    S.synth_state_dict(blk, "winosite.")
    x = K.nchw_to_nhwc(_rand("wsx", (n, c, h, h)).to(dev))
    emb = blk.local_embed(_rand("wse", (n, 64)).to(dev))
    old = BLK.WINOGRAD
    lib = L.load()
    try:
        BLK.WINOGRAD = 2
        counts, outs = [], []
        for k in range(3):
            handle = C.c_void_p()
            L.check(lib.mf_cmdlist_begin(), "begin")
            outs.append(blk(x, emb).clone())
            L.check(lib.mf_cmdlist_end(C.byref(handle)), "end")
            counts.append(lib.mf_cmdlist_count(handle))
            lib.mf_cmdlist_free(handle)
    finally:
        BLK.WINOGRAD = old
    site = blk.block_seq[0].basic_block.conv._wino_sites
    assert site and all(site.values()), site
    assert torch.equal(outs[0], outs[1]) and torch.equal(outs[1], outs[2])
    # the first evaluation also packs weights and runs the stand-alone transforms (of x and of block 0's output); later ones: embedding bound (2),
    # GEMM + tail of each of the two convolutions (4) -- block 0's output comes with its transform-domain mirror, x keeps its cached one
    assert counts[1] == counts[2] == 6 and counts[0] > counts[1], counts
    print(f"[measured] launches of a UnetResBlock on the Winograd form: first evaluation {counts[0]}, later {counts[1]}")


@pytest.mark.parametrize("shape", [(16, 8, 8, 512, 0, 1024), (16, 8, 8, 1024, 1024, 1024), (16, 8, 8, 1024, 512, 512), (16, 16, 16, 512, 512, 512), (16, 16, 16, 512, 256, 256)])
def test_conv_res_in_the_grid_of_the_component_gemm(dev, shape):
    """conv_res (1x1) of a channel-changing ResBlock as the GUEST of its 3x3's Winograd component GEMM (mf_conv2d_wino_gn_apply_f16x2(..., guest)):
    the same bits as conv_res in a launch of its own, one launch fewer"""
    import ctypes as C
    from medfusion_amd import blocks as BLK, kernels as K, lib as L
    n, h, w, c1, c2, co = shape
    blk = BLK.BasicResBlock(2, c1 + c2, co, 3, 1, ("GROUP", {"num_groups": 32, "affine": True}), ("Swish", {})).to(dev)
    S.synth_state_dict(blk, f"winogrp{shape}.")
    x1 = K.nchw_to_nhwc(_rand(f"wgx{shape}", (n, c1, h, w)).to(dev))
    x2 = K.nchw_to_nhwc(_rand(f"wgy{shape}", (n, c2, h, w)).to(dev)) if c2 else None
    x = x1 if x2 is None else (x1, x2)
    lib = L.load()
    old = BLK.WINOGRAD, BLK.WINO_GROUP
    outs, counts = {}, {}
    try:
        BLK.WINOGRAD = 2
        for grp in (False, True):
            BLK.WINO_GROUP = grp
            blk(x)                                  # (weights packed, mirrors cached)
            handle = C.c_void_p()
            L.check(lib.mf_cmdlist_begin(), "begin")
            outs[grp] = blk(x).clone()
            L.check(lib.mf_cmdlist_end(C.byref(handle)), "end")
            counts[grp] = lib.mf_cmdlist_count(handle)
            lib.mf_cmdlist_free(handle)
        assert blk._wino_guest(x) is not None, shape
    finally:
        BLK.WINOGRAD, BLK.WINO_GROUP = old
    assert torch.equal(outs[True], outs[False]), shape
    assert counts[True] == counts[False] - 1 == 2, counts     # GEMM (+ conv_res in its grid) and the tail


F32_TAIL_CASES = [
    # (N, H, W, C1, C2, Cout, G, residual, emb, affine, act, precision)
    (16, 8, 8, 1024, 0, 1024, 32, True, True, True, 1, 3),      # published 8 x 8 level, exact bf16 triplets
    (16, 8, 8, 1024, 512, 512, 32, True, True, True, 1, 3),     # two-source out block
    (16, 16, 16, 512, 0, 512, 32, True, False, True, 1, 3),     # published 16 x 16 level
    (4, 16, 16, 64, 0, 128, 8, False, False, False, 0, 3),      # no affine, no act, no residual
    (2, 32, 32, 64, 0, 256, 32, True, True, True, 1, 3),
    (16, 8, 8, 512, 0, 1024, 32, True, True, True, 1, 0),       # the same pieces on the fp32 MFMA kernel
]


@pytest.mark.parametrize("case", F32_TAIL_CASES, ids=lambda c: "x".join(str(v) for v in c))
def test_conv_wino_gn_apply_on_the_exact_arithmetics(dev, case):
    """Round 6 (ABI 250): the Winograd form on MF_CONV_FP32_SPLIT3_W3 / MF_CONV_FP32 -- fp32 input transform, the 16 component GEMMs on the exact
    arithmetic's own implicit-GEMM kernel (upsample = 3 descriptor), the tail kernel of the fp16-pair form in its fp32 mode -- against (a) the DIRECT
    convolution of the same arithmetic + the GroupNorm apply pass and (b) the chain in fp64; its transform-domain output equals the stand-alone fp32
    transform of its own output bit for bit."""
    from medfusion_amd import kernels as K, lib as L
    n, h, w, c1, c2, co, G, has_res, has_emb, affine, act, prec = case
    cin = c1 + c2
    x = _rand(f"fx{case}", (n, c1, h, w))
    x2 = _rand(f"fy{case}", (n, c2, h, w)) if c2 else None
    wt = _rand(f"fw{case}", (co, cin, 3, 3), 1.0 / np.sqrt(cin * 9))
    b = _rand(f"fb{case}", (co,), 0.1)
    gamma = (1.0 + 0.3 * _rand(f"fg{case}", (co,))) if affine else None
    beta = 0.2 * _rand(f"fbe{case}", (co,)) if affine else None
    res = _rand(f"fr{case}", (n, co, h, w), 2.0) if has_res else None
    emb = _rand(f"fe{case}", (n, co), 0.5) if has_emb else None
    xd = K.nchw_to_nhwc(x.to(dev))
    x2d = K.nchw_to_nhwc(x2.to(dev)) if c2 else None
    resd = K.nchw_to_nhwc(res.to(dev)) if has_res else None
    embd = emb.to(dev) if has_emb else None
    gd, bd = (gamma.to(dev), beta.to(dev)) if affine else (None, None)
    d = K.make_conv_desc(n, h, w, c1, c2, co, 3, 1, 1, 0, precision=prec)
    assert K.wino_f32_ok(d, G), case
    u = K.wino_pack_weight(wt.to(dev))
    u = K.split_conv_weight(u) if prec == 3 else u
    got = K.conv2d_wino_gn_apply_f32(xd, u, b.to(dev), d, gd, bd, G, 1e-5, act=act, residual=resd, emb=embd, emb_stride=embd.stride(0) if has_emb else 0, x2=x2d,
                                     want_wino=True)
    # (a) the direct form of the same arithmetic
    wp = K.pack_conv_weight(wt.to(dev))
    wp = K.split_conv_weight(wp) if prec == 3 else wp
    y = K.conv2d(xd, wp, b.to(dev), d, x2=x2d)
    ref = K.gn_apply(y, K.gn_stats(y, G, 1e-5), gd, bd, G, act, resd, embd, embd.stride(0) if has_emb else 0)
    e_direct = relerr(got, ref)
    # (b) fp64
    xin = x if x2 is None else torch.cat([x, x2], 1)
    y64 = F.conv2d(xin.double(), wt.double(), b.double(), padding=1)
    t64 = F.group_norm(y64, G, gamma.double() if affine else None, beta.double() if affine else None, 1e-5)
    if act:
        t64 = t64 * torch.sigmoid(t64)
    if has_res:
        t64 = t64 + res.double()
    if has_emb:
        t64 = t64 + emb.double()[:, :, None, None]
    e64 = relerr(K.nhwc_to_nchw(got), t64)
    e64d = relerr(K.nhwc_to_nchw(ref), t64)
    print(f"[measured] winograd on the exact arithmetic {case}: vs fp64 chain {e64:.2e} (direct form of the same arithmetic: {e64d:.2e}); vs that direct form {e_direct:.2e}")
    assert e64 < 5e-6 and e_direct < 5e-6, (case, e64, e_direct)
    v = got._mf_wino_f32
    got2 = got.clone()
    assert torch.equal(K.wino_input_f32(got2), v), case
    # V against an fp64 transform of the output
    T = (h // 2) * (w // 2)
    g64 = F.pad(K.nhwc_to_nchw(got).cpu().double(), (1, 1, 1, 1))
    patches = g64.unfold(2, 4, 2).unfold(3, 4, 2)                       # [n, c, h/2, w/2, 4, 4]
    v64 = torch.einsum("ij,nchwjk,lk->ilnhwc", BT, patches, BT).reshape(16, n, T, co)
    assert relerr(v, v64) < 1e-6, case


def test_wino_f32_refuses_what_it_cannot_do(dev):
    from medfusion_amd import kernels as K
    assert not K.wino_f32_ok(K.make_conv_desc(16, 8, 8, 1024, 0, 1024, 3, 1, 1, 0, precision=5), 32)     # the pair arithmetic has its own entry points
    assert not K.wino_f32_ok(K.make_conv_desc(16, 8, 8, 1024, 0, 1024, 3, 2, 1, 0, precision=3), 32)     # stride 2
    assert not K.wino_f32_ok(K.make_conv_desc(3, 8, 8, 1024, 0, 1024, 3, 1, 1, 0, precision=3), 32)      # 48 tile rows per component: no tile
    assert not K.wino_f32_ok(K.make_conv_desc(2, 64, 64, 64, 0, 512, 3, 1, 1, 0, precision=3), 32)       # 64 x 64 x 16 channels do not fit in LDS
    assert K.wino_f32_ok(K.make_conv_desc(16, 16, 16, 512, 0, 512, 3, 1, 1, 0, precision=3), 32)
