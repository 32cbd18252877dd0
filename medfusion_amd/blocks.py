"""Building blocks of the UNet / VAE on the HIP kernels (host-side mirror of the reference's
medical_diffusion/models/utils/{conv_blocks,attention_blocks}.py: same class names, constructor
meaning and state-dict keys; spatial_dims=2 only).

Modules hold their parameters in the reference layout (OIHW conv weights) so reference checkpoints
load with `load_state_dict`; device-side packed copies ([Cout][KH][KW][Cin]) are built lazily and
rebuilt when a parameter changes.  `forward` takes NHWC tensors [N,H,W,C] on the GPU.  An input may be
a pair (h, skip): the channel concat of unet2.py:259 is fused into the convolutions that consume it.
There is no CPU path.
"""
from __future__ import annotations

import os
from typing import Optional, Tuple, Union

import torch
import torch.nn as nn

from . import kernels as K
from . import lib as L

Act = Union[torch.Tensor, Tuple[torch.Tensor, torch.Tensor]]


def monai_padding(kernel_size: int, stride: int) -> int:
    """MONAI get_padding as used at conv_blocks.py:48,169,229: int((k - s + 1) / 2)."""
    p = (kernel_size - stride + 1) / 2
    if p < 0:
        raise AssertionError("padding value should not be negative")
    return int(p)


def zero_module(m: nn.Module) -> nn.Module:
    """attention_blocks.py:27-33"""
    for p in m.parameters():
        p.detach().zero_()
    return m


def _split(x: Act):
    return (x[0], x[1]) if isinstance(x, (tuple, list)) else (x, None)


class _Packed:
    """Lazily packed device copy of a conv weight, invalidated by in-place parameter updates."""

    def __init__(self, subpixel: bool = False, pad_cin: int = 0):
        self._pad_cin = pad_cin
        self._key = None
        self._w = None
        self._w3 = None
        self._wb = None
        self._wh = None
        self._wu = None
        self._l1 = {}
        self._subpixel = subpixel

    def get(self, weight: torch.Tensor) -> torch.Tensor:
        key = (weight.data_ptr(), weight._version, weight.device)
        if key != self._key:
            w = weight.detach()
            if w.dim() == 3:  # Conv1d [O, I, 1]
                w = w.unsqueeze(-1)
            if self._pad_cin and w.shape[1] < self._pad_cin:   # zero input channels up to pad_cin (load-time plumbing): the padded pair operand's zeros meet zeros
                w = torch.nn.functional.pad(w, (0, 0, 0, 0, 0, self._pad_cin - w.shape[1]))
            self._w = K.pack_upconv_weight(w) if self._subpixel else K.pack_conv_weight(w)
            self._w3 = None
            self._wb = None
            self._wh = None
            self._wu = None
            self._l1 = {}
            self._raw = w
            self._key = key
        return self._w

    def get_wino(self, weight: torch.Tensor):
        """the Winograd-domain weights U = G g G^T of a 3x3 convolution as fp16 pairs (mf_wino_pack_weight_f32, then split like any weight)"""
        self.get(weight)
        if self._wu is None:
            self._wu = K.split_weight_f16x2(K.wino_pack_weight(self._raw))
        return self._wu

    def get_wino_f32(self, weight: torch.Tensor, split3: bool):
        """U = G g G^T [16, Cout, 1, 1, Cin] for the Winograd form on the exact arithmetics: plain fp32 (MF_CONV_FP32) or the bf16 triplets of
        MF_CONV_FP32_SPLIT3_W3 (rows = 16 Cout); derived once per weight version"""
        self.get(weight)
        if getattr(self, "_wuf_key", None) != self._key:
            self._wuf, self._wu3, self._wuf_key = K.wino_pack_weight(self._raw), None, self._key
        if not split3:
            return self._wuf
        if self._wu3 is None:
            self._wu3 = K.split_conv_weight(self._wuf)
        return self._wu3

    def get_f16x2(self, weight: torch.Tensor) -> torch.Tensor:
        """the same weights as fp16 pairs (MF_CONV_FP32_F16X2), derived once from the fp32 packing"""
        wp = self.get(weight)
        if self._wh is None:
            self._wh = K.split_weight_f16x2(wp)   # (pairs scaled by max |w|, that max)
        return self._wh

    def l1(self, weight: torch.Tensor, c1: int):
        """(largest L1 norm of a filter over input channels [0, c1), ... over [c1, Cin)) of the packed weights: the operand side of the
        DERIVED output bound of mf_conv2d_f16x2_pairs_out (|y| <= bound(x1) l1[0] + bound(x2) l1[1] + max |bias|).  Load-time host sync, cached."""
        wp = self.get(weight)
        ent = self._l1.get(c1)
        if ent is None:
            a = wp[..., :c1].abs().sum(dim=(-3, -2, -1)).max()
            b = wp[..., c1:].abs().sum(dim=(-3, -2, -1)).max() if c1 < wp.shape[-1] else torch.zeros((), device=wp.device)
            # (a hair above the fp32 sums: they feed an UPPER bound)
            ent = self._l1[c1] = (float(a.item()) * (1.0 + 1e-5), float(b.item()) * (1.0 + 1e-5))
        return ent

    def get_bf16(self, weight: torch.Tensor) -> torch.Tensor:
        """the same weights rounded to bf16 (opt-in MF_CONV_BF16)"""
        wp = self.get(weight)
        if self._wb is None:
            self._wb = K.convert_conv_weight_bf16(wp)
        return self._wb

    def get_split(self, weight: torch.Tensor) -> torch.Tensor:
        """the same weights as bf16 triplets (MF_CONV_FP32_SPLIT3_W3), derived once from the fp32 packing"""
        wp = self.get(weight)
        if self._w3 is None:
            self._w3 = K.split_conv_weight(wp)
        return self._w3


PAIRS_ONLY_BETWEEN_BLOCKS = os.environ.get("MEDFUSION_PAIRS_ONLY", "1") != "0"   # (A/B switch of the pairs-only apply output)
# the NCHW input convolution on the fp16-pair kernel (mf_pack_nchw_pairs_f32 + zero-padded weights): built, tested, and measured NOT faster on cfg2
# (33.22 / 33.24 vs 33.23 / 33.32 images/s, two interleaved rounds: the 9-chunk matrix launch + the pack launch cost what the fp32 direct kernel
# + the measuring / split passes cost) -- opt-in, MEDFUSION_INPUT_CONV_PAIRS=1
INPUT_CONV_ON_PAIRS = os.environ.get("MEDFUSION_INPUT_CONV_PAIRS", "0") == "1"
# conv_res in the launch of the block's 3x3 (mf_conv2d_f16x2_group; conv_f16x2_group.h).  Bit-identical to the two launches and +2.2 % on the cfg2
# step (same-process A/B, four interleaved rounds: 454.8 -> 444.9 ms, profiles/r04_grouped_conv_res_ab.txt).  MEDFUSION_GROUPED_CONV_RES=0: two launches (A/B)
GROUPED_CONV_RES = os.environ.get("MEDFUSION_GROUPED_CONV_RES", "1") != "0"
GROUP_GUEST = {}   # tuning hook (scripts/group_tune.py): (N, H, W, C1, C2, Cout) -> guest tile (or (tile, split-K)) for that block, -1 = two launches; empty in the product
DERIVED_OUT_BOUNDS = os.environ.get("MEDFUSION_DERIVED_BOUNDS", "1") != "0"   # (A/B switch of mf_conv2d_f16x2_pairs_out behind down / up convolutions)
SUBPIXEL_UPSAMPLE = True  # BasicUp as the sub-pixel (transposed-conv-equivalent) form whenever the shape allows
# Arithmetic of the implicit-GEMM convolutions (include/medfusion_hip.h, MF_CONV_*); read per call: set blocks.CONV_PRECISION or the env var.
#   5 (default) fp32 through PAIRS of fp16: 23-bit operands with a per-sample power-of-two scale, three product terms on the fp16 matrix
#     cores, both operands moved to LDS by LDS-DMA -- error vs fp64 BELOW the fp32-MFMA kernel's (profiles/r02_split_accuracy.txt), the
#     150-iteration trajectory at 1e-6 of the oracle, 1.5x the speed of 1.  Convolutions that are not on that kernel (edge layers) run as 1.
#   1 fp32 operands split EXACTLY into three bf16 terms (24 bits; weights split once at load), six product terms on the bf16 matrix cores;
#   0 v_mfma_f32_32x32x2_f32 (bit-for-bit an fp32 fma chain);
#   4 opt-in REDUCED precision (operands rounded to bf16, one MFMA term; its own tolerance) -- never a default.
#   6 opt-in REDUCED precision on the LDS-DMA kernel of 5: the same fp16-pair operands, ONE product term (operands rounded to fp16, 11 bits;
#     MF_CONV_F16) -- its own tolerance, never a default, never the headline.
CONV_PRECISION = int(os.environ.get("MEDFUSION_CONV_PRECISION", "5"))
# Winograd F(2x2, 3x3) form of the 3x3 stride-1 convolutions (arithmetic 5 only; kernels.conv2d_wino_f16x2): 0 never, 1 (default) where the library
# prefers it (mf_wino_preferred: a rule fitted to the MI355X sweeps that holds at any batch, round 6; it admits every shape of csrc/wino_plan_table.inc),
# 2 wherever the library can (tests, sweeps).  Read per call like CONV_PRECISION.
WINOGRAD = int(os.environ.get("MEDFUSION_WINOGRAD", "1"))


# the same form on the EXACT arithmetics (CONV_PRECISION 1 = bf16 triplets; round 6, VERDICT r05 Next #9): fp32 transforms, the 16 component GEMMs on the exact
# arithmetic's own kernel, the same tail kernel with fp32 outputs.  0 never (the direct form: what `other_conv_arithmetic` reported until round 5), 1 (default) on
# the shapes the fp16-pair rule admits and every shape up to 16 x 16, 2 wherever the library can.  CONV_PRECISION 0 (the bit-for-bit fp32 MFMA chain) always stays on the direct form.
WINOGRAD_F32 = int(os.environ.get("MEDFUSION_WINOGRAD_F32", "1"))
# ... and, opt-in only, on CONV_PRECISION 0: the plain fp32 MFMA kernel runs the component GEMMs (no operand splitting anywhere: fp32 transforms, fp32 matrix
# instructions, fp32 tail) -- bench.py times it next to the direct chain; the default for CONV_PRECISION 0 stays the bit-for-bit direct form
WINOGRAD_F32_MFMA = int(os.environ.get("MEDFUSION_WINOGRAD_F32_MFMA", "0"))
WINO_TAIL = os.environ.get("MEDFUSION_WINOGRAD_TAIL", "1") != "0"   # the GroupNorm / Swish / residual / embedding tail in the launch behind the GEMM (A/B switch; 0: three launches)
WINO_GROUP = os.environ.get("MEDFUSION_WINOGRAD_GROUP", "1") != "0"  # conv_res in the grid of its ResBlock's component GEMM (A/B switch; 0: its own launch)
WINO_SHAPES = {}   # tuning hook (scripts/wino_sweep.py / wino_ab.py): (N, H, W, Cin, Cout) -> (tile, split-K) of the component GEMM (0: planner); empty in the product
if os.environ.get("MEDFUSION_WINOGRAD_TABLE"):   # a JSON list of [N, H, W, Cin, Cout, tile, split-K] (what the sweep writes), for A/B runs without a rebuild
    import json as _json
    WINO_SHAPES = {tuple(e[:5]): (int(e[5]), int(e[6])) for e in _json.load(open(os.environ["MEDFUSION_WINOGRAD_TABLE"]))}


def wino_wanted(d) -> bool:
    if WINOGRAD == 0:
        return False
    if (d.N, d.Hin, d.Win, d.C1 + d.C2, d.Cout) in WINO_SHAPES:
        return K.wino_ok(d)
    return K.wino_ok(d) if WINOGRAD == 2 else K.wino_preferred(d)


def f16x2_mode() -> bool:
    """the activations travel with fp16-pair mirrors (arithmetics 5 and 6 read them)"""
    return CONV_PRECISION in (5, 6)


class Conv(nn.Module):
    """Parameter holder + launcher for one convolution (nn.Conv2d replacement, never calls ATen)."""

    def __init__(self, in_ch, out_ch, kernel_size, stride=1, padding=0, upsample=False, conv1d=False):
        super().__init__()
        holder = nn.Conv1d(in_ch, out_ch, 1) if conv1d else nn.Conv2d(in_ch, out_ch, kernel_size, stride, padding, bias=True)
        self.weight, self.bias = holder.weight, holder.bias  # default torch init, reference key names
        self.in_ch, self.out_ch, self.k, self.stride, self.pad, self.upsample = in_ch, out_ch, kernel_size, stride, padding, int(upsample)
        self._packed = _Packed()
        self._packed_sub = _Packed(subpixel=True)
        self._packed_pad = _Packed(pad_cin=32)
        self._pad_ok = {}
        self._descs = {}
        self._pairs_out = {}
        self._wino_sites = {}     # shape key -> a consumer transformed this output into the Winograd domain: write V in the tail from now on
        self._bmax = (None, 0.0)

    def _forward_f16x2(self, x1, x2, n, h, w, c1, c2, out, gn_groups, gn_eps, measure_out, prec=5, derive_out=False, cin_pad=0):
        """MF_CONV_FP32_F16X2 (prec 5) / MF_CONV_F16 (prec 6), or None when this convolution is not on that kernel.
        cin_pad: the weights zero-padded to that many input channels (the input convolution on a padded pair operand)"""
        key = ("f16x2", n, h, w, c1, c2, gn_groups, prec, WINOGRAD)
        ent = self._descs.get(key)
        if ent is None:
            d = K.make_conv_desc(n, h, w, c1, c2, self.out_ch, self.k, self.stride, self.pad, 2 if self.upsample else 0, precision=prec)
            ok = K.conv_f16x2_ok(d)
            pinned = K.pin_conv_plan(d) if ok else None     # (tile, split-K) fixed in the descriptor: per-launch planning is a field read
            wino = None
            if ok and prec == 5 and not cin_pad and self.k == 3 and wino_wanted(d):
                # the Winograd form is the faster one for this shape (mf_wino_preferred): 2.25x fewer matrix instructions
                wparts = K.wino_gn_parts(d, gn_groups) if gn_groups else 0
                if not gn_groups or wparts > 0:
                    wt, wsk = WINO_SHAPES.get((n, h, w, c1 + c2, self.out_ch), (0, 0))
                    wd = K.make_conv_desc(n, h, w, c1, c2, self.out_ch, self.k, self.stride, self.pad, 0, tile_hint=wt, splitk_hint=wsk, precision=prec)   # (hints address the component GEMM)
                    wino = (wd, wparts, K.pin_wino_plan(wd))
            ent = (d, K.conv_gn_parts(d, gn_groups) if (ok and gn_groups) else 0, ok, pinned, wino)
            self._descs[key] = ent
        d, parts, ok, pinned, wino = ent
        if not ok:
            return None
        if wino is not None and out is None and not measure_out and not derive_out:
            wd, wparts, wpinned = wino
            r = K.conv2d_wino_f16x2(x1, self._packed.get_wino(self.weight), self.bias, wd, x2=x2, gn_groups=gn_groups, gn_parts=wparts, pinned=wpinned)
            return (r[0], K.GnPartials(r[1], wparts, gn_eps)) if gn_groups else r
        pk = self._packed_pad if cin_pad else (self._packed_sub if d.upsample == 2 else self._packed)
        wh = pk.get_f16x2(self.weight)
        if not gn_groups:
            if derive_out and out is None and DERIVED_OUT_BOUNDS:
                # an output that feeds convolutions un-normalised (down- / up-sampling): its fp16-pair form straight from the epilogue, under
                # a bound derived from the operands -- no measuring pass, no split launch in front of the consumers.  ONLY where the bound
                # does not propagate: a derived bound sits ~2^11 above the true maximum, harmless for one hop (the pair format keeps its
                # 23 bits down to 2^-28 of the bound) but it compounds -- derived from derived along the residual stream (conv_res ->
                # apply -> conv_res ...) it reached 2^44 x the data after seven blocks and the pairs lost their bits (scripts/debug_derived.py,
                # round 4).  The consumers of a down- / up-sampled tensor are a GroupNorm'd convolution and a conv_res that MEASURES: both reset.
                po = self._pairs_out.get(key)
                if po is None:
                    po = self._pairs_out[key] = K.conv_pairs_out_ok(d)
                if po:
                    bkey = (self.bias._version, self.bias.data_ptr())
                    if self._bmax[0] != bkey:
                        self._bmax = (bkey, float(self.bias.detach().abs().max().item()))
                    return K.conv2d_f16x2_pairs_out(x1, wh, self.bias, d, pk.l1(self.weight, c1), self._bmax[1], x2=x2, pinned=pinned)
            return K.conv2d_f16x2(x1, wh, self.bias, d, x2=x2, out=out, measure_out=measure_out, pinned=pinned)
        ho, wo = K.conv_out_hw(d)
        if parts > 0:
            y, partial = K.conv2d_f16x2(x1, wh, self.bias, d, x2=x2, out=out, gn_groups=gn_groups, gn_parts=parts, pinned=pinned)
        else:
            y = K.conv2d_f16x2(x1, wh, self.bias, d, x2=x2, out=out, pinned=pinned)
            partial, parts = K.gn_stats_partial(y, gn_groups)
        # the apply pass reduces the records itself (mf_gn_apply_from_partials_f32: 30.5 vs 30.3 images/s against a finalize launch per norm)
        if gn_groups <= 256:
            return y, K.GnPartials(partial, parts, gn_eps)
        return y, K.gn_finalize(partial, parts, ho * wo, self.out_ch, gn_groups, gn_eps)   # (more groups than a workgroup has threads)

    def wino_tail_desc(self, x: Act, G: int):
        """(descriptor, pinned sizes) of this convolution on the Winograd form with its GroupNorm tail for the input x, or None; cached per shape"""
        x1, x2 = _split(x)
        n, h, w, c1 = x1.shape
        c2 = 0 if x2 is None else x2.shape[-1]
        key = ("wino_tail", n, h, w, c1, c2, G, WINOGRAD, WINO_TAIL)
        ent = self._descs.get(key)
        if ent is None:
            ent = False
            if WINO_TAIL and CONV_PRECISION == 5 and self.k == 3 and self.stride == 1 and not self.upsample and c1 + c2 == self.in_ch:
                d = K.make_conv_desc(n, h, w, c1, c2, self.out_ch, self.k, self.stride, self.pad, 0, precision=5)
                if wino_wanted(d) and K.wino_tail_ok(d, G):
                    wt, wsk = WINO_SHAPES.get((n, h, w, c1 + c2, self.out_ch), (0, 0))
                    d.tile_hint, d.splitk_hint = wt, wsk
                    ent = (d, K.pin_wino_plan(d))
            self._descs[key] = ent
        return (key, ent) if ent is not False else (key, None)

    def forward_wino_gn_apply(self, x: Act, norm, act: int, residual, emb, emb_stride, out_fp32, bconst, guest=None):
        """conv -> GroupNorm -> Swish -> + residual -> + emb on the Winograd form, the tail in ONE launch behind the component GEMM
        (mf_conv2d_wino_gn_apply_f16x2), or None when this convolution is not on that path for this shape.  The output carries its transform-domain
        mirror as well once a Winograd convolution has asked for it (K.wino_input marks the site): the next call writes it in the tail."""
        x1, x2 = _split(x)
        n, h, w, c1 = x1.shape
        c2 = 0 if x2 is None else x2.shape[-1]
        if c1 + c2 != self.in_ch:
            raise RuntimeError(f"conv expects {self.in_ch} input channels, got {c1}+{c2}")
        G = norm.num_groups
        key, ent = self.wino_tail_desc(x, G)
        if ent is None:
            return None
        d, pinned = ent
        want = self._wino_sites.get(key, False)
        y = K.conv2d_wino_gn_apply(x1, self._packed.get_wino(self.weight), self.bias, d, norm.weight, norm.bias, G, norm.eps, act=act, residual=residual,
                                   emb=emb, emb_stride=emb_stride, x2=x2, bconst=bconst, out_fp32=out_fp32, want_wino=want, pinned=pinned, guest=guest)
        if not want:
            y._mf_wino_site = (self._wino_sites, key)
        return y

    def forward_wino_gn_apply_f32(self, x: Act, norm, act: int, residual, emb, emb_stride, split3: bool = True):
        """conv -> GroupNorm -> Swish -> + residual -> + emb on the Winograd form of the exact bf16-triplet arithmetic (kernels.conv2d_wino_gn_apply_f32), or
        None when this convolution is not on that path for this shape.  Like the fp16-pair form, the output carries V for the next Winograd
        convolution once one has asked for it (the site flag is learnt on the eager first iteration)."""
        x1, x2 = _split(x)
        n, h, w, c1 = x1.shape
        c2 = 0 if x2 is None else x2.shape[-1]
        G = norm.num_groups
        mode = WINOGRAD_F32 if split3 else WINOGRAD_F32_MFMA
        key = ("wino_f32", n, h, w, c1, c2, G, mode, split3)
        ent = self._descs.get(key)
        if ent is None:
            ent = False
            if self.k == 3 and self.stride == 1 and not self.upsample and c1 + c2 == self.in_ch:
                d = K.make_conv_desc(n, h, w, c1, c2, self.out_ch, 3, 1, 1, 0, precision=3 if split3 else 0)
                if K.wino_f32_ok(d, G) and K.conv_is_igemm(d):
                    rule = K.make_conv_desc(n, h, w, c1, c2, self.out_ch, 3, 1, 1, 0, precision=5)   # (the shape rule of mf_wino_preferred speaks of the pair arithmetic's descriptor)
                    # (1: the pair arithmetic's rule, plus every shape up to 16 x 16 -- with six matrix terms per product the low-channel shapes the pair
                    # arithmetic leaves direct gain here too: +0.95 % on the step, profiles/r06_small_ab.txt section 7; the 32 x 32 level loses: mode 2)
                    if mode == 2 or h * w <= 256 or K.wino_preferred(rule):
                        ent = d
            self._descs[key] = ent
        if ent is False:
            return None
        K._need_f32(x1, x2, residual)
        want = self._wino_sites.get(key, False)
        y = K.conv2d_wino_gn_apply_f32(x1, self._packed.get_wino_f32(self.weight, split3), self.bias, ent, norm.weight, norm.bias, G, norm.eps, act=act,
                                       residual=residual, emb=emb, emb_stride=emb_stride, x2=x2, want_wino=want)
        if not want:
            y._mf_wino_site_f32 = (self._wino_sites, key)
        return y

    def forward_gn_apply(self, x: Act, norm, act: int, residual, emb, emb_stride, out_fp32, bconst):
        """conv -> GroupNorm -> Swish -> + residual -> + emb in ONE launch (mf_conv2d_f16x2_gn_apply), or None when this convolution cannot
        (not on the fp16-pair kernel, or a plan whose workgroups are not all resident at once): the caller takes the two-launch form"""
        x1, x2 = _split(x)
        n, h, w, c1 = x1.shape
        c2 = 0 if x2 is None else x2.shape[-1]
        if c1 + c2 != self.in_ch:
            raise RuntimeError(f"conv expects {self.in_ch} input channels, got {c1}+{c2}")
        prec = CONV_PRECISION
        G = norm.num_groups
        key = ("fused", n, h, w, c1, c2, G, prec)
        ent = self._descs.get(key)
        if ent is None:
            d = K.make_conv_desc(n, h, w, c1, c2, self.out_ch, self.k, self.stride, self.pad, 2 if self.upsample else 0, precision=prec)
            ent = (None, 0, 0, None)
            ho, wo = K.conv_out_hw(d)
            if K.conv_f16x2_ok(d) and not self.upsample and ho * wo >= K.FUSE_MIN_HW:
                pinned = K.pin_conv_plan(d)
                parts, words = K.conv_gn_parts(d, G), K.conv_fuse_words(d, G)
                if parts > 0 and words > 0:
                    ent = (d, parts, words, pinned)
            self._descs[key] = ent
        d, parts, words, pinned = ent
        if d is None or K.Rendezvous.disabled:
            return None
        return K.conv2d_f16x2_gn_apply(x1, self._packed.get_f16x2(self.weight), self.bias, d, norm.weight, norm.bias, G, norm.eps, parts, words, act=act,
                                       residual=residual, emb=emb, emb_stride=emb_stride, x2=x2, bconst=bconst, out_fp32=out_fp32, pinned=pinned)

    def forward(self, x: Act, in_layout=L.LAYOUT_NHWC, out_layout=L.LAYOUT_NHWC, out=None, rows: Optional[slice] = None, gn_groups: int = 0,
                gn_eps: float = 1e-5, measure_out: bool = False, derive_out: bool = False):
        """gn_groups > 0: also return the statistics of the GroupNorm that follows -> (y, stats [N,G,2]).
        measure_out: in the fp16-pair mode, also measure the per-sample max |y| (y feeds a convolution or a residual add un-normalised).
        derive_out: ... or, where the plan allows, write y's fp16-pair form in the epilogue under a bound DERIVED from the operands (see
        _forward_f16x2; only for tensors whose bound does not propagate: the outputs of down- / up-sampling)."""
        x1, x2 = _split(x)
        if in_layout == L.LAYOUT_NCHW:
            n, c1, h, w = x1.shape
        else:
            n, h, w, c1 = x1.shape
        c2 = 0 if x2 is None else x2.shape[-1]
        if c1 + c2 != self.in_ch:
            raise RuntimeError(f"conv expects {self.in_ch} input channels, got {c1}+{c2}")
        prec = CONV_PRECISION
        if prec not in (0, 1, 4, 5, 6):
            raise RuntimeError(f"blocks.CONV_PRECISION = {prec}: 0 (fp32 MFMA), 1 (exact bf16 triplets), 5 (fp16 pairs, default), or the opt-in "
                               f"reduced precisions 4 (bf16) / 6 (fp16)")
        if prec in (5, 6):
            if rows is None and in_layout == L.LAYOUT_NHWC and out_layout == L.LAYOUT_NHWC:
                r = self._forward_f16x2(x1, x2, n, h, w, c1, c2, out, gn_groups, gn_eps, measure_out, prec, derive_out)
                if r is not None:
                    return r
            if (INPUT_CONV_ON_PAIRS and rows is None and in_layout == L.LAYOUT_NCHW and out_layout == L.LAYOUT_NHWC and x2 is None and out is None
                    and not gn_groups and not self.upsample and c1 < 32 and self.out_ch % 64 == 0 and c1 * h * w <= (1 << 18)
                    and self._pad_ok.get((n, h, w, prec), True)):
                # the network's input convolution (8 -> 256 at the UNet, 3 -> 64 at the VAE encoder): the NCHW input goes to a 32-channel
                # fp16-pair operand in one small launch (measuring its own bound), the weights are zero-padded to 32 input channels once, and
                # the convolution runs on the matrix cores, writing its output's pair form under a derived bound -- instead of the fp32
                # direct kernel + a measuring pass + a split pass (27 + 5 + 8 us per iteration -> 4 + 14)
                if (n, h, w, prec) not in self._pad_ok:   # (asked once per shape: is the padded convolution on the pair kernel at all?)
                    self._pad_ok[(n, h, w, prec)] = K.conv_f16x2_ok(K.make_conv_desc(n, h, w, 32, 0, self.out_ch, self.k, self.stride, self.pad, 0, precision=prec))
                if self._pad_ok[(n, h, w, prec)]:
                    r = self._forward_f16x2(K.pack_nchw_pairs(x1, 32), None, n, h, w, 32, 0, None, 0, gn_eps, True, prec, True, cin_pad=32)
                    if r is not None:
                        return r
            prec = 1  # not on the fp16-pair kernel (edge convolutions, odd channel counts): the exact bf16-triplet / plain fp32 kernels
            K._need_f32(x1, x2)
        key = (n, h, w, c1, c2, in_layout, out_layout, rows.start if rows else None, gn_groups, prec)
        ent = self._descs.get(key)
        cout = self.out_ch if rows is None else rows.stop - rows.start
        if ent is None:
            dp = 3 if prec == 1 else prec   # exact bf16 triplets = MF_CONV_FP32_SPLIT3_W3 (the weights are split once at load)
            d = K.make_conv_desc(n, h, w, c1, c2, cout, self.k, self.stride, self.pad, self.upsample, in_layout, out_layout, precision=dp)
            if self.upsample and SUBPIXEL_UPSAMPLE and rows is None:
                d2 = K.make_conv_desc(n, h, w, c1, c2, cout, self.k, self.stride, self.pad, 2, in_layout, out_layout, precision=dp)
                if K.subpixel_ok(d2):  # 4 phase-specific 2x2 convs on the low-res tensor: 4/9 of the MACs
                    d = d2
            if d.precision in (3, 4) and (rows is not None or not K.conv_is_igemm(d)):
                d.precision = 0  # the small / edge convolutions are not on the implicit-GEMM kernel (and a row slice has no converted weights): plain fp32
            ent = (d, K.conv_gn_parts(d, gn_groups) if gn_groups else 0)
            self._descs[key] = ent
        d, parts = ent
        pk = self._packed_sub if d.upsample == 2 else self._packed
        wp = pk.get_split(self.weight) if d.precision == 3 else pk.get_bf16(self.weight) if d.precision == 4 else pk.get(self.weight)
        b = self.bias
        if rows is not None:  # output-channel slice (learned-variance head split)
            wp, b = wp[rows], b[rows]
        if not gn_groups:
            return K.conv2d(x1, wp, b, d, x2=x2, out=out)
        ho, wo = K.conv_out_hw(d)
        if parts > 0:    # statistics fused into the conv epilogue / split-K reducer; finalize: a separate tiny kernel (the fused
            # last-arriver and in-apply variants of round 1 measured slower and are gone)
            y, partial = K.conv2d_gn(x1, wp, b, d, gn_groups, parts, x2=x2)
        else:
            y = K.conv2d(x1, wp, b, d, x2=x2, out=out)
            partial, parts = K.gn_stats_partial(y, gn_groups)
        return y, K.gn_finalize(partial, parts, ho * wo, cout, gn_groups, gn_eps)


class GroupNorm(nn.Module):
    """nn.GroupNorm parameter holder (keys weight/bias)."""

    def __init__(self, num_groups, num_channels, eps=1e-5, affine=True):
        super().__init__()
        self.num_groups, self.num_channels, self.eps = num_groups, num_channels, eps
        if affine:
            self.weight = nn.Parameter(torch.ones(num_channels))
            self.bias = nn.Parameter(torch.zeros(num_channels))
        else:
            self.weight = self.bias = None
        self._bkey, self._bmax = None, (1.0, 0.0)

    def bound_const(self, group_size: int) -> float:
        """upper bound of |gn(x) * gamma + beta| (and of its Swish): a normalised group of n values has |value| <= sqrt(n - 1)"""
        if self.weight is not None:
            key = (self.weight._version, self.bias._version, self.weight.data_ptr())
            if key != self._bkey:  # load-time host sync, cached
                self._bmax = (float(self.weight.detach().abs().max().item()), float(self.bias.detach().abs().max().item()))
                self._bkey = key
        return self._bmax[0] * float(group_size) ** 0.5 + self._bmax[1]


def _norm(norm_name, channels) -> GroupNorm:
    kind, kw = norm_name
    if kind.upper() != "GROUP":
        raise NotImplementedError(f"norm {kind}: only GROUP is on the sampling path")
    return GroupNorm(num_channels=channels, **kw)


class BasicBlock(nn.Module):
    """conv -> GroupNorm -> (Dropout: identity at inference) -> Swish.  conv_blocks.py:134-192."""

    def __init__(self, spatial_dims, in_channels, out_channels, kernel_size, stride=1, norm_name=None, act_name=None, dropout=None,
                 zero_conv=False):
        super().__init__()
        assert spatial_dims == 2
        conv = Conv(in_channels, out_channels, kernel_size, stride, monai_padding(kernel_size, stride))
        self.conv = zero_module(conv) if zero_conv else conv
        if norm_name is not None:
            self.norm = _norm(norm_name, out_channels)
        self.has_act = act_name is not None

    def forward(self, x: Act, residual=None, emb=None, emb_stride=0, in_layout=L.LAYOUT_NHWC, out_layout=L.LAYOUT_NHWC, out_fp32=True, wino_guest=None):
        """wino_guest: conv_res of the enclosing ResBlock for the launch of this block's Winograd component GEMM (BasicResBlock._wino_guest)"""
        has_norm = hasattr(self, "norm")
        if has_norm:
            if out_layout != L.LAYOUT_NHWC:
                raise RuntimeError("norm/act epilogue needs NHWC")
            if CONV_PRECISION == 5 and WINOGRAD and WINO_TAIL and in_layout == L.LAYOUT_NHWC:
                nm = self.norm
                x1 = _split(x)[0]
                y = self.conv.forward_wino_gn_apply(x, nm, int(self.has_act), residual, emb, emb_stride, out_fp32,
                                                    nm.bound_const(x1.shape[1] * x1.shape[2] * (self.conv.out_ch // nm.num_groups)), guest=wino_guest)
                if y is not None:
                    return y
            if wino_guest is not None:
                raise RuntimeError("BasicBlock: a guest convolution was planned for a Winograd launch that did not happen")
            if ((CONV_PRECISION == 1 and WINOGRAD_F32) or (CONV_PRECISION == 0 and WINOGRAD_F32_MFMA)) and in_layout == L.LAYOUT_NHWC and not isinstance(residual, (tuple, list)):
                y = self.conv.forward_wino_gn_apply_f32(x, self.norm, int(self.has_act), residual, emb, emb_stride, split3=CONV_PRECISION == 1)
                if y is not None:
                    return y
            if f16x2_mode() and in_layout == L.LAYOUT_NHWC and not K.Rendezvous.disabled:
                # one launch for conv + GroupNorm + Swish + residual + embedding where the plan allows it (conv_f16x2.h: FuseP)
                nm = self.norm
                x1 = _split(x)[0]
                ho, wo = (x1.shape[1] + 2 * self.conv.pad - self.conv.k) // self.conv.stride + 1, (x1.shape[2] + 2 * self.conv.pad - self.conv.k) // self.conv.stride + 1
                bc = nm.bound_const(ho * wo * (self.conv.out_ch // nm.num_groups))
                y = self.conv.forward_gn_apply(x, nm, int(self.has_act), residual, emb, emb_stride, out_fp32, bc)
                if y is not None:
                    return y
            return self.finish(self.conv_and_stats(x, in_layout), residual, emb, emb_stride, out_fp32)
        y = self.conv(x, in_layout=in_layout, out_layout=out_layout)
        if not (self.has_act or residual is not None or emb is not None):
            return y
        if out_layout != L.LAYOUT_NHWC:
            raise RuntimeError("act/residual epilogue needs NHWC")
        return K.gn_apply(y, None, None, None, 1, int(self.has_act), residual, emb, emb_stride, out=y)


def _basicblock_conv_and_stats(self, x, in_layout=L.LAYOUT_NHWC):
    """conv (+ fused GroupNorm statistics) -> (y, stats)"""
    return self.conv(x, in_layout=in_layout, gn_groups=self.norm.num_groups, gn_eps=self.norm.eps)


def _basicblock_finish(self, y_stats, residual=None, emb=None, emb_stride=0, out_fp32=True):
    y, stats = y_stats
    nm = self.norm
    split = f16x2_mode()
    bc = nm.bound_const(y.shape[1] * y.shape[2] * (y.shape[3] // nm.num_groups)) if split else 0.0
    return K.gn_apply(y, stats, nm.weight, nm.bias, nm.num_groups, int(self.has_act), residual, emb, emb_stride, out=y, split=split, bconst=bc,
                      out_fp32=out_fp32)


BasicBlock.conv_and_stats = _basicblock_conv_and_stats
BasicBlock.finish = _basicblock_finish


class BasicResBlock(nn.Module):
    """BasicBlock(x) + (conv1x1(x) if Cin != Cout else x).  conv_blocks.py:194-240.
    The norm/Swish/residual-add/embedding-add are one fused pass (mf_gn_apply_f32)."""

    def __init__(self, spatial_dims, in_channels, out_channels, kernel_size, stride=1, norm_name=None, act_name=None, dropout=None,
                 zero_conv=False):
        super().__init__()
        self.basic_block = BasicBlock(spatial_dims, in_channels, out_channels, kernel_size, stride, norm_name, act_name, dropout, zero_conv)
        self.conv_res = Conv(in_channels, out_channels, 1, stride, monai_padding(1, stride)) if in_channels != out_channels else nn.Identity()
        self._pairs_ok = {}
        self._group = {}

    def forward(self, x: Act, emb=None, emb_stride=0, in_layout=L.LAYOUT_NHWC, out_fp32=True):
        """out_fp32=False: the caller promises that the block's output is read by fp16-pair convolutions and residual adds only"""
        if isinstance(self.conv_res, nn.Identity):
            if isinstance(x, (tuple, list)) or in_layout != L.LAYOUT_NHWC:
                raise RuntimeError("identity residual needs a single NHWC input")
            return self.basic_block(x, residual=x, emb=emb, emb_stride=emb_stride, in_layout=in_layout, out_fp32=out_fp32)
        if GROUPED_CONV_RES and CONV_PRECISION == 5 and in_layout == L.LAYOUT_NHWC and hasattr(self.basic_block, "norm") and K.Rendezvous.disabled:
            g = self._grouped(x)
            if g is not None:
                # conv_res and the block's 3x3 read the same x and not each other: ONE launch (mf_conv2d_f16x2_group) -- the 3x3's workgroups
                # first, the 1x1's on the CUs they leave; bit-identical to the two launches below
                x1, x2 = _split(x)
                bb, nm = self.basic_block, self.basic_block.norm
                (y, partial), res = K.conv2d_f16x2_group(
                    x1, x2,
                    dict(w_split=bb.conv._packed.get_f16x2(bb.conv.weight), bias=bb.conv.bias, d=g[0], gn_groups=nm.num_groups, gn_parts=g[1], pinned=g[2]),
                    dict(w_split=self.conv_res._packed.get_f16x2(self.conv_res.weight), bias=self.conv_res.bias, d=g[3], pinned=g[4]))
                return bb.finish((y, K.GnPartials(partial, g[1], nm.eps)), res, emb, emb_stride, out_fp32)
        if WINO_GROUP and GROUPED_CONV_RES and CONV_PRECISION == 5 and WINOGRAD and WINO_TAIL and in_layout == L.LAYOUT_NHWC and hasattr(self.basic_block, "norm"):
            g = self._wino_guest(x)
            if g is not None:
                # the 3x3 runs as Winograd component GEMMs: conv_res joins THAT launch (mf_conv2d_wino_gn_apply_f16x2(..., guest)); the tail behind
                # it adds conv_res's output as the residual -- bit-identical to the separate conv_res launch
                cr = self.conv_res
                return self.basic_block(x, residual=None, emb=emb, emb_stride=emb_stride, in_layout=in_layout, out_fp32=out_fp32,
                                        wino_guest=dict(w_split=cr._packed.get_f16x2(cr.weight), bias=cr.bias, d=g[0], pinned=g[1]))
        res = self.conv_res(x, in_layout=in_layout, measure_out=f16x2_mode())
        return self.basic_block(x, residual=res, emb=emb, emb_stride=emb_stride, in_layout=in_layout, out_fp32=out_fp32)

    def _wino_guest(self, x):
        """(descriptor, pinned) of conv_res as the guest of the 3x3's Winograd component GEMM for this input shape, or None; cached per shape.
        The guest keeps the split-K of the plan it has alone (the summation order: same bits as its own launch), on an 8-wave tile."""
        x1, x2 = _split(x)
        n, h, w, c1 = x1.shape
        c2 = 0 if x2 is None else x2.shape[-1]
        key = ("wino", n, h, w, c1, c2, WINOGRAD, WINO_TAIL)
        if key in self._group:
            return self._group[key]
        ent = None
        c3, cr, nm = self.basic_block.conv, self.conv_res, self.basic_block.norm
        _, tail = c3.wino_tail_desc(x, nm.num_groups) if (c1 + c2 == c3.in_ch) else (None, None)
        if tail is not None and cr.k == 1 and cr.stride == 1:
            nat = K.make_conv_desc(n, h, w, c1, c2, cr.out_ch, cr.k, cr.stride, cr.pad, 0, precision=5)
            sk0 = K.conv_plan(nat)[1] if K.conv_f16x2_ok(nat) else 0
            m = n * h * w
            t36, t37 = -(-m // 128) * (cr.out_ch // 64), -(-m // 64) * max(cr.out_ch // 256, 1)
            order = (36, 37) if abs(t36 - 256) <= abs(t37 - 256) or cr.out_ch % 256 else (37, 36)
            for tile in (*order, 53):      # (53: the 4-wave guest of a 4-wave host tile -- the round-6 GEMM model picks tile 54 for 1536 -> 512 at 8 x 8)
                if (tile == 37 and cr.out_ch % 256) or sk0 <= 0:
                    continue
                db = K.make_conv_desc(n, h, w, c1, c2, cr.out_ch, cr.k, cr.stride, cr.pad, 0, tile_hint=tile, splitk_hint=sk0, precision=5)
                if K.conv_f16x2_ok(db) and K.wino_group_ok(tail[0], db):
                    pb = K.pin_conv_plan(db)
                    if pb[1] > 0:      # (its output is measured: the tail reads the slot maxima)
                        ent = (db, pb)
                        break
        self._group[key] = ent
        return ent

    def _grouped(self, x):
        """(descriptor, GroupNorm parts, pinned) of the 3x3 + (descriptor, pinned) of conv_res when the two can share a launch for this input
        shape, else None; cached per shape.  conv_res keeps the planner's tile when its workgroup size equals the 3x3's, otherwise the first
        of the guest tiles of that size the pair is instantiated for (descriptor hints)."""
        x1, x2 = _split(x)
        n, h, w, c1 = x1.shape
        c2 = 0 if x2 is None else x2.shape[-1]
        key = (n, h, w, c1, c2, WINOGRAD, WINO_TAIL)
        if key in self._group:
            return self._group[key]
        ent = None
        c3, cr, nm = self.basic_block.conv, self.conv_res, self.basic_block.norm
        G = nm.num_groups
        if c1 + c2 == c3.in_ch and not c3.upsample and G <= 256:
            da = K.make_conv_desc(n, h, w, c1, c2, c3.out_ch, c3.k, c3.stride, c3.pad, 0, precision=5)
            # (a 3x3 that WILL run as Winograd component GEMM + tail has its own shared launch, _wino_guest; asked of the path that is actually taken --
            # wino_tail_desc -- not of wino_wanted alone: a shape the tail cannot take, or MEDFUSION_WINOGRAD_TAIL=0, keeps this grouped launch.  ADVICE r05)
            takes_wino_tail = WINO_TAIL and c3.wino_tail_desc(x, G)[1] is not None
            if K.conv_f16x2_ok(da) and not takes_wino_tail:
                pa = K.pin_conv_plan(da)
                parts = K.conv_gn_parts(da, G)
                # guest tiles of the 8-wave hosts: the one whose grid is closest to one workgroup per CU first (profiles/r04_conv_sweep_planner_vs_best.txt)
                m = n * ((h + 2 * cr.pad - cr.k) // cr.stride + 1) * ((w + 2 * cr.pad - cr.k) // cr.stride + 1)
                t36, t37 = -(-m // 128) * (cr.out_ch // 64), -(-m // 64) * max(cr.out_ch // 256, 1)
                wide = (36, 37) if abs(t36 - 256) <= abs(t37 - 256) or cr.out_ch % 256 else (37, 36)
                # (the guest keeps the SPLIT-K of the plan it has alone: the tile only partitions pixels and output channels, but the number of
                # K slices is the summation order -- with it fixed, the block gives the same bits whether or not the pair shares a launch)
                nat = K.make_conv_desc(n, h, w, c1, c2, cr.out_ch, cr.k, cr.stride, cr.pad, 0, precision=5)
                sk0 = K.conv_plan(nat)[1] if K.conv_f16x2_ok(nat) else 0
                forced = GROUP_GUEST.get((n, h, w, c1, c2, cr.out_ch))
                if isinstance(forced, tuple):      # (tile, split-K): the tuner may also try another summation order (same value to fp32 rounding)
                    forced, sk0 = forced
                for tile in ((0, *wide, 53) if forced is None else () if forced < 0 else (forced,)):
                    if (tile == 37 and cr.out_ch % 256) or sk0 <= 0:
                        continue
                    db = K.make_conv_desc(n, h, w, c1, c2, cr.out_ch, cr.k, cr.stride, cr.pad, 0, tile_hint=tile, splitk_hint=sk0 if tile else 0, precision=5)
                    if not K.conv_f16x2_ok(db):
                        continue
                    pb = K.pin_conv_plan(db)
                    if parts > 0 and pb[1] > 0 and K.conv_group_ok(da, G, db, 0):
                        ent = (da, parts, pa, db, pb)
                        break
        self._group[key] = ent
        return ent

    def reads_pairs_only(self, n: int, h: int, w: int, c2: int = 0) -> bool:
        """can this block take an [n, h, w, Cin] input (c2 > 0: a fused concat whose second source has c2 channels) that exists as fp16 pairs only?
        Its convolutions must be on the fp16-pair kernel FOR THAT SHAPE (its residual add reads pairs anyway): asked of the planner itself
        (mf_conv2d_f16x2_ok on the real descriptor -- tensor size limits, tile fits -- not guessed from channel counts: ADVICE r03), cached per shape"""
        if not (f16x2_mode() and hasattr(self.basic_block, "norm")):
            return False
        key = (n, h, w, c2, CONV_PRECISION)
        ok = self._pairs_ok.get(key)
        if ok is None:
            convs = [self.basic_block.conv] + ([] if isinstance(self.conv_res, nn.Identity) else [self.conv_res])
            ok = all(c.in_ch > c2 and K.conv_f16x2_ok(K.make_conv_desc(n, h, w, c.in_ch - c2, c2, c.out_ch, c.k, c.stride, c.pad, 0, precision=CONV_PRECISION)) for c in convs)
            self._pairs_ok[key] = ok
        return ok


class _EmbBlock(nn.Module):
    BlockCls = None
    emb_after_last = False  # UnetBasicBlock adds emb after every block (conv_blocks.py:300), UnetResBlock not after the last (:362)

    def __init__(self, spatial_dims, in_channels, out_channels, kernel_size, stride=1, norm_name=None, act_name=None, dropout=None,
                 emb_channels=None, blocks=2):
        super().__init__()
        self.out_channels = out_channels
        self.block_seq = nn.ModuleList([
            self.BlockCls(spatial_dims, in_channels if i == 0 else out_channels, out_channels, kernel_size, stride, norm_name, act_name,
                          dropout, i == blocks - 1)
            for i in range(blocks)])
        if emb_channels is not None:
            # Swish -> Linear(emb_channels, out_channels); index 1 carries the parameters (key `local_embedder.1.*`)
            self.local_embedder = nn.Sequential(nn.Identity(), nn.Linear(emb_channels, out_channels))

    def forward(self, x: Act, emb: Optional[torch.Tensor] = None, in_layout=L.LAYOUT_NHWC, out_fp32: bool = True):
        """`emb`: the block's *local* embedding [B, Cout] (already through Swish->Linear; the UNet batches
        all local embedders into one GEMM), possibly a strided view into a wider matrix.
        out_fp32=False: the caller promises that the block's OUTPUT is read by fp16-pair convolutions and residual adds only (UNet.features)."""
        n = len(self.block_seq)
        last = n if self.emb_after_last else n - 1
        for i, blk in enumerate(self.block_seq):
            e = emb if (emb is not None and i < last) else None
            es = e.stride(0) if e is not None else 0
            if isinstance(blk, BasicResBlock):
                # the output of every block but the last is read by the NEXT block alone: its convolutions (fp16 pairs) and its residual
                # add -- the fp32 form need not be written when that block can read pairs (12 instead of 16 bytes per element)
                nxt = self.block_seq[i + 1] if i + 1 < n else None
                pairs_next = False
                lay = in_layout if i == 0 else L.LAYOUT_NHWC
                if PAIRS_ONLY_BETWEEN_BLOCKS and nxt is not None:
                    x1 = _split(x)[0]
                    nn_, hh, ww = (x1.shape[0], x1.shape[2], x1.shape[3]) if lay == L.LAYOUT_NCHW else (x1.shape[0], x1.shape[1], x1.shape[2])
                    c0 = blk.basic_block.conv
                    ho, wo = (hh + 2 * c0.pad - c0.k) // c0.stride + 1, (ww + 2 * c0.pad - c0.k) // c0.stride + 1   # shape of THIS block's output
                    pairs_next = nxt.reads_pairs_only(nn_, ho, wo)
                x = blk(x, emb=e, emb_stride=es, in_layout=lay, out_fp32=(not pairs_next) if nxt is not None else out_fp32)
            else:
                x = blk(x, emb=e, emb_stride=es, in_layout=in_layout if i == 0 else L.LAYOUT_NHWC)
        return x

    def local_embed(self, emb: torch.Tensor) -> torch.Tensor:
        """Stand-alone local embedding (used when the block is driven outside a UNet)."""
        lin = self.local_embedder[1]
        return K.linear(emb, lin.weight, lin.bias, act_in=True)


class UnetResBlock(_EmbBlock):
    """conv_blocks.py:305-364"""
    BlockCls = BasicResBlock
    emb_after_last = False


class UnetBasicBlock(_EmbBlock):
    """conv_blocks.py:244-302"""
    BlockCls = BasicBlock
    emb_after_last = True


class BasicDown(nn.Module):
    """conv_blocks.py:28-70.  Learnable: 3x3 stride-s conv (key `down_op.*`), plus -- `use_res` -- nn.PixelUnshuffle(2)(x) added to its output
    (:54-55,68-69; needs out_channels == 4 in_channels; no model of the reference sets it, built for completeness);
    learnable_interpolation=False: nn.AvgPool2d(k, stride, get_padding(k, stride)), no parameters, the channel count stays."""

    def __init__(self, spatial_dims, in_channels, out_channels, kernel_size=3, stride=2, learnable_interpolation=True, use_res=False):
        super().__init__()
        self.learnable = bool(learnable_interpolation)
        self.use_res = bool(use_res) and self.learnable        # (the reference creates `down_skip` inside the learnable branch only)
        # the output's fp16-pair form under a bound DERIVED from the operands -- unless its consumer adds it as a RESIDUAL (the VAE's DownBlock /
        # UpBlock: an identity-residual ResBlock follows): there the derived bound, ~2^12-2^20 above the data, would travel on through every
        # apply pass of the level (bound = bconst + bound(residual)); that tensor is measured instead (round 5, the bound-slack audit)
        self.derive_out = True
        if self.learnable:
            self.down_op = Conv(in_channels, out_channels, kernel_size, stride, monai_padding(kernel_size, stride))
        else:
            self.k, self.stride, self.pad = kernel_size, stride, monai_padding(kernel_size, stride)

    def forward(self, x, emb=None):
        if self.learnable:
            if not self.use_res:
                return self.down_op(x, measure_out=f16x2_mode(), derive_out=self.derive_out)
            if isinstance(x, (tuple, list)):
                raise RuntimeError("BasicDown(use_res=True) takes one tensor")
            return K.pixel_unshuffle2_add(x, self.down_op(x))   # (the sum is measured by its first fp16-pair consumer)
        if isinstance(x, (tuple, list)):
            raise RuntimeError("BasicDown(learnable_interpolation=False) takes one tensor")
        return K.avgpool2d(x, self.k, self.stride, self.pad)


class BasicUp(nn.Module):
    """conv_blocks.py:72-131: nearest-exact x2 then 3x3 conv, fused into one gather (key `up_op.*`), plus -- `use_res` -- nn.PixelShuffle(2)(x)
    added to its output (:114-115,125-126; out_channels == in_channels / 4); learnable_interpolation=False: the plain nearest-exact resize
    (:128-130), no parameters."""

    def __init__(self, spatial_dims, in_channels, out_channels, kernel_size=2, stride=2, learnable_interpolation=True, use_res=False):
        super().__init__()
        if (kernel_size, stride) != (2, 2):
            raise NotImplementedError("BasicUp: only x2 upsampling (kernel_size=stride=2) is supported")
        self.learnable = bool(learnable_interpolation)
        self.use_res = bool(use_res) and self.learnable
        self.derive_out = True     # (see BasicDown)
        if self.learnable:
            self.up_op = Conv(in_channels, out_channels, 3, 1, 1, upsample=True)

    def forward(self, x, emb=None):
        if self.learnable:
            if not self.use_res:
                return self.up_op(x, measure_out=f16x2_mode(), derive_out=self.derive_out)
            if isinstance(x, (tuple, list)):
                raise RuntimeError("BasicUp(use_res=True) takes one tensor")
            return K.pixel_shuffle2_add(x, self.up_op(x))
        if isinstance(x, (tuple, list)):
            raise RuntimeError("BasicUp(learnable_interpolation=False) takes one tensor")
        return K.upsample_nearest2x(x)


class SequentialEmb(nn.Sequential):
    """conv_blocks.py:21-25 (emb is a dict here: per-module local embeddings keyed by module id)."""

    def forward(self, x, emb_lookup, out_fp32: bool = True):
        """out_fp32=False (UNet.features): whoever reads this sequence's output reads fp16 pairs only -- forwarded to its last conv block when
        nothing behind it in the sequence needs fp32 (an Attention that is the identity; a learnable BasicUp: a fp16-pair convolution)"""
        mods = list(self)
        pairs = set()
        if not out_fp32:   # (the caller has checked that every conv block of the sequence can read pairs: UNet._pairs_only_outputs)
            for i, m in enumerate(mods):
                if isinstance(m, _EmbBlock):
                    j = i + 1
                    while j < len(mods) and not isinstance(mods[j], _EmbBlock):
                        j += 1
                    if all((isinstance(t, Attention) and not hasattr(t, "attention")) or (isinstance(t, BasicUp) and t.learnable and not t.use_res)
                           for t in mods[i + 1:j]):
                        pairs.add(id(m))
        for m in mods:
            x = m(x, emb_lookup(m), out_fp32=False) if id(m) in pairs else m(x, emb_lookup(m))
        return x


# ----------------------------------------------------------------------------- attention (optional path)
class LinearTransformer(nn.Module):
    """attention_blocks.py:128-195.  With an embedding the cross-attention has ONE key, so softmax == 1 and
    out = x + to_out(to_v(emb)) broadcast over space -- computed in that exact closed form."""

    def __init__(self, spatial_dims, in_channels, out_channels, num_heads, ch_per_head=32, norm_name=("GROUP", {"num_groups": 32, "affine": True}),
                 dropout=None, emb_dim=None):
        super().__init__()
        hid = num_heads * ch_per_head
        self.num_heads, self.scale, self.hid = num_heads, ch_per_head ** -0.25, hid
        self.in_channels, self.out_channels = in_channels, out_channels
        self.norm_x = _norm(norm_name, in_channels)
        self.cross = emb_dim is not None
        emb_dim = in_channels if emb_dim is None else emb_dim
        self.to_q = Conv(in_channels, hid, 1, conv1d=True)
        self.to_k = Conv(emb_dim, hid, 1, conv1d=True)
        self.to_v = Conv(emb_dim, hid, 1, conv1d=True)
        self.to_out = nn.Sequential(zero_module(Conv(hid, out_channels, 1, conv1d=True)), nn.Identity())

    def forward(self, x, embedding=None):
        n, h, w, c = x.shape
        if embedding is not None:
            v = K.linear(embedding, self.to_v.weight.view(self.hid, -1), self.to_v.bias)          # [B, hid] (one key)
            o = K.linear(v, self.to_out[0].weight.view(self.out_channels, -1), self.to_out[0].bias)  # [B, Cout]
            if self.out_channels != c:
                raise NotImplementedError("LinearTransformer with out_channels != in_channels")
            return K.gn_apply(x, None, None, None, 1, 0, None, o, o.stride(0))
        stats = K.gn_stats(x, self.norm_x.num_groups, self.norm_x.eps)
        x_n = K.gn_apply(x, stats, self.norm_x.weight, self.norm_x.bias, self.norm_x.num_groups, 0)
        q, k, v = self.to_q(x_n), self.to_k(x_n), self.to_v(x_n)
        a = K.attention(q.view(n, h * w, self.hid), k.view(n, h * w, self.hid), v.view(n, h * w, self.hid), self.num_heads, self.scale)
        out = self.to_out[0](a.view(n, h, w, self.hid))
        return K.add(x, out, out=out) if out.shape == x.shape else out


class GEGLU(nn.Module):
    """attention_blocks.py:11-25"""

    def __init__(self, in_channels, out_channels):
        super().__init__()
        self.norm = nn.LayerNorm(in_channels)
        self.proj = nn.Linear(in_channels, out_channels * 2, bias=True)
        self._packed = _Packed()
        self.out_channels = out_channels

    def forward(self, x):
        n, h, w, c = x.shape
        xn = K.layernorm(x, self.norm.weight, self.norm.bias, self.norm.eps)
        d = K.make_conv_desc(n, h, w, c, 0, 2 * self.out_channels, 1, 1, 0)
        hgate = K.conv2d(xn, self._packed.get(self.proj.weight.view(2 * self.out_channels, c, 1, 1)), self.proj.bias, d)
        return K.geglu(hgate)


class BasicTransformerBlock(nn.Module):
    """attention_blocks.py:200-231"""

    def __init__(self, spatial_dims, in_channels, out_channels, num_heads, ch_per_head=32, norm_name=("GROUP", {"num_groups": 32, "affine": True}),
                 dropout=None, emb_dim=None):
        super().__init__()
        self.self_atn = LinearTransformer(spatial_dims, in_channels, in_channels, num_heads, ch_per_head, norm_name, dropout, None)
        if emb_dim is not None:
            self.cros_atn = LinearTransformer(spatial_dims, in_channels, in_channels, num_heads, ch_per_head, norm_name, dropout, emb_dim)
        self.proj_out = nn.Sequential(GEGLU(in_channels, in_channels * 4), nn.Identity(), Conv(in_channels * 4, out_channels, 1))

    def forward(self, x, embedding=None):
        x = self.self_atn(x)
        if embedding is not None:
            x = self.cros_atn(x, embedding=embedding)
        out = self.proj_out[2](self.proj_out[0](x))
        return K.add(out, x, out=out) if out.shape[-1] == x.shape[-1] else x


class SpatialTransformer(nn.Module):
    """attention_blocks.py:233-288"""

    def __init__(self, spatial_dims, in_channels, out_channels, num_heads, ch_per_head=32, norm_name=("GROUP", {"num_groups": 32, "affine": True}),
                 dropout=None, emb_dim=None, depth=1):
        super().__init__()
        self.norm = _norm(norm_name, in_channels)
        hid = num_heads * ch_per_head
        self.proj_in = Conv(in_channels, hid, 1)
        self.transformer_blocks = nn.ModuleList([
            BasicTransformerBlock(spatial_dims, hid, hid, num_heads, ch_per_head, norm_name, dropout=dropout, emb_dim=emb_dim) for _ in range(depth)])
        self.proj_out = Conv(hid, out_channels, 1)

    def forward(self, x, embedding=None):
        stats = K.gn_stats(x, self.norm.num_groups, self.norm.eps)
        h = K.gn_apply(x, stats, self.norm.weight, self.norm.bias, self.norm.num_groups, 0)
        h = self.proj_in(h)
        for blk in self.transformer_blocks:
            h = blk(h, embedding=embedding)
        h = self.proj_out(h)
        return K.add(h, x, out=h) if h.shape == x.shape else h


class Attention(nn.Module):
    """attention_blocks.py:291-335: 'none' -> identity, 'linear', 'spatial'."""

    def __init__(self, spatial_dims, in_channels, out_channels, num_heads=8, ch_per_head=32, norm_name=("GROUP", {"num_groups": 32, "affine": True}),
                 dropout=0, emb_dim=None, depth=1, attention_type="linear"):
        super().__init__()
        if attention_type == "spatial":
            self.attention = SpatialTransformer(spatial_dims, in_channels, out_channels, num_heads, ch_per_head, norm_name, dropout, emb_dim, depth)
        elif attention_type == "linear":
            self.attention = LinearTransformer(spatial_dims, in_channels, out_channels, num_heads, ch_per_head, norm_name, dropout, emb_dim)

    def forward(self, x, emb=None):
        """`emb` here is the GLOBAL embedding [B, emb_dim] (time + condition), as in the reference."""
        return self.attention(x, emb) if hasattr(self, "attention") else x


class DownBlock(nn.Module):
    """VAE encoder stage (conv_blocks.py:368-441)"""

    def __init__(self, spatial_dims, in_channels, out_channels, kernel_size, stride, downsample_kernel_size, norm_name, act_name, dropout=None,
                 use_res_block=False, learnable_interpolation=True, use_attention="none", emb_channels=None):
        super().__init__()
        enable_down = stride != 1
        down_out = out_channels if learnable_interpolation and enable_down else in_channels
        self.down_op = BasicDown(spatial_dims, in_channels, out_channels, downsample_kernel_size, stride, learnable_interpolation) if enable_down else nn.Identity()
        self.attention = Attention(spatial_dims, down_out, down_out, 8, down_out // 8, norm_name, dropout, emb_channels, 1, use_attention)
        Blk = UnetResBlock if use_res_block else UnetBasicBlock
        self.conv_block = Blk(spatial_dims, down_out, out_channels, kernel_size, 1, norm_name, act_name, dropout, emb_channels)
        if enable_down and use_res_block and down_out == out_channels:
            self.down_op.derive_out = False     # its output is the identity residual of conv_block's first ResBlock: measured, not derived

    def forward(self, x, emb=None):
        x = self.down_op(x)
        x = self.attention(x, emb)
        return self.conv_block(x, None)


class UpBlock(nn.Module):
    """VAE decoder stage (conv_blocks.py:444-528)"""

    def __init__(self, spatial_dims, in_channels, out_channels, kernel_size, stride, upsample_kernel_size, norm_name, act_name, dropout=None,
                 use_res_block=False, learnable_interpolation=True, use_attention="none", emb_channels=None, skip_channels=0):
        super().__init__()
        enable_up = stride != 1
        skip_out = out_channels if learnable_interpolation and enable_up else in_channels + skip_channels
        self.learnable_interpolation = bool(learnable_interpolation)
        self.up_op = BasicUp(spatial_dims, in_channels, out_channels, upsample_kernel_size, stride, learnable_interpolation) if enable_up else nn.Identity()
        self.attention = Attention(spatial_dims, skip_out, skip_out, 8, skip_out // 8, norm_name, dropout, emb_channels, 1, use_attention)
        Blk = UnetResBlock if use_res_block else UnetBasicBlock
        self.conv_block = Blk(spatial_dims, skip_out, out_channels, kernel_size, 1, norm_name, act_name, dropout, emb_channels)
        if enable_up and use_res_block and skip_out == out_channels:
            self.up_op.derive_out = False       # (see DownBlock)

    def forward(self, x_enc, x_skip=None, emb=None):
        x = self.up_op(x_enc)
        if x_skip is not None:
            if self.learnable_interpolation:     # conv_blocks.py:516-519: equal channel counts -> sum, else concatenate
                x = K.add(x, x_skip, out=x)
            else:
                if hasattr(self.attention, "attention"):
                    raise NotImplementedError("UpBlock(learnable_interpolation=False) with a skip AND attention")
                return self.conv_block((x, x_skip), None)   # the concat is fused into the consuming convolutions
        x = self.attention(x, emb)
        return self.conv_block(x, None)
