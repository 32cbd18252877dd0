"""ORACLE -- TEST INFRASTRUCTURE ONLY.

CPU restatement (plain PyTorch fp32, no MONAI / Lightning / streamlit) of the reference's
latent-diffusion *sampling* path.  Only `tests/`, `__graft_entry__.smoke()` and the
`cpu_baseline` leg of `bench.py` may import this module; the product (`medfusion_amd`)
never does.  Every class cites the reference file:line it restates (paths relative to
/root/reference).  State-dict keys are identical to the reference's so one weight set can
drive the reference (build container), this oracle, and the HIP path.

Parity pin: `oracle/gen_golden.py` imports the real reference here (with `oracle/shims`)
and checks this restatement is *bit-identical* to it on CPU for every fixture written to
`tests/golden/` (the reference's own tests hold no numeric vectors -- SURVEY.md §4).
"""
from __future__ import annotations

import math
from typing import Callable, Optional, Sequence

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F


# ----------------------------------------------------------------------------- helpers
def monai_padding(kernel_size: int, stride: int) -> int:
    """MONAI `get_padding` (call sites conv_blocks.py:48,169,229): int((k - s + 1) / 2)."""
    p = (kernel_size - stride + 1) / 2
    if p < 0:
        raise AssertionError("negative padding")
    return int(p)


class Swish(nn.Module):
    """MONAI Swish, alpha=1: `x * sigmoid(1.0 * x)` (not F.silu) -- SURVEY Q12."""

    def forward(self, x):
        return x * torch.sigmoid(1.0 * x)


def zero_module(m: nn.Module) -> nn.Module:
    """attention_blocks.py:27-33."""
    for p in m.parameters():
        p.detach().zero_()
    return m


def save_add(*args):
    """conv_blocks.py:16-18."""
    args = [a for a in args if a is not None]
    return sum(args) if len(args) > 0 else None


def _norm(norm_name, channels):
    kind, kw = norm_name
    assert kind.upper() == "GROUP"
    return nn.GroupNorm(num_channels=channels, **kw)


# ----------------------------------------------------------------------------- conv blocks
class BasicBlock(nn.Module):
    """conv -> GroupNorm -> (Dropout, identity in eval) -> Swish.  conv_blocks.py:134-192."""

    def __init__(self, in_ch, out_ch, kernel_size, stride=1, norm_name=None, act=False, zero_conv=False):
        super().__init__()
        conv = nn.Conv2d(in_ch, out_ch, kernel_size, stride, monai_padding(kernel_size, stride), bias=True)
        self.conv = zero_module(conv) if zero_conv else conv
        if norm_name is not None:
            self.norm = _norm(norm_name, out_ch)
        if act:
            self.act = Swish()

    def forward(self, x):
        out = self.conv(x)
        if hasattr(self, "norm"):
            out = self.norm(out)
        if hasattr(self, "act"):
            out = self.act(out)
        return out


class BasicResBlock(nn.Module):
    """BasicBlock(x) + (conv1x1(x) if Cin != Cout else x).  conv_blocks.py:194-240."""

    def __init__(self, in_ch, out_ch, kernel_size, stride=1, norm_name=None, act=False, zero_conv=False):
        super().__init__()
        self.basic_block = BasicBlock(in_ch, out_ch, kernel_size, stride, norm_name, act, zero_conv)
        self.conv_res = (
            nn.Conv2d(in_ch, out_ch, 1, stride, monai_padding(1, stride), bias=True) if in_ch != out_ch else nn.Identity()
        )

    def forward(self, x):
        out = self.basic_block(x)
        return out + self.conv_res(x)


class _EmbBlock(nn.Module):
    """Shared body of UnetResBlock / UnetBasicBlock (conv_blocks.py:244-364).

    `emb_after_last`: UnetBasicBlock adds the embedding after *every* block (i < n, :300),
    UnetResBlock after all but the last (i < n-1, :362) -- SURVEY Q13.
    """

    BlockCls = None
    emb_after_last = False

    def __init__(self, in_ch, out_ch, kernel_size, stride=1, norm_name=None, act=True, emb_channels=None, blocks=2):
        super().__init__()
        self.block_seq = nn.ModuleList(
            [
                self.BlockCls(in_ch if i == 0 else out_ch, out_ch, kernel_size, stride, norm_name, act, i == blocks - 1)
                for i in range(blocks)
            ]
        )
        if emb_channels is not None:
            self.local_embedder = nn.Sequential(Swish(), nn.Linear(emb_channels, out_ch))

    def forward(self, x, emb=None):
        if emb is not None:
            emb = self.local_embedder(emb)
            b, c = emb.shape[:2]
            emb = emb.reshape(b, c, 1, 1)
        n = len(self.block_seq)
        last = n if self.emb_after_last else n - 1
        for i, blk in enumerate(self.block_seq):
            x = blk(x)
            if emb is not None and i < last:
                x += emb  # in place, as the reference
        return x


class UnetResBlock(_EmbBlock):
    BlockCls = BasicResBlock
    emb_after_last = False


class UnetBasicBlock(_EmbBlock):
    BlockCls = BasicBlock
    emb_after_last = True


class SequentialEmb(nn.Sequential):
    """conv_blocks.py:21-25."""

    def forward(self, x, emb):
        for m in self:
            x = m(x, emb)
        return x


class BasicDown(nn.Module):
    """Learnable: 3x3 stride-s conv (+ nn.PixelUnshuffle(2) skip with use_res, :54-55,68-69); else AvgPool.  conv_blocks.py:28-70."""

    def __init__(self, in_ch, out_ch, kernel_size=3, stride=2, learnable_interpolation=True, use_res=False):
        super().__init__()
        if learnable_interpolation:
            self.down_op = nn.Conv2d(in_ch, out_ch, kernel_size, stride, monai_padding(kernel_size, stride), bias=True)
            if use_res:
                self.down_skip = nn.PixelUnshuffle(2)
        else:
            self.down_op = nn.AvgPool2d(kernel_size, stride, monai_padding(kernel_size, stride))

    def forward(self, x, emb=None):
        y = self.down_op(x)
        if hasattr(self, "down_skip"):
            y = y + self.down_skip(x)
        return y


class BasicUp(nn.Module):
    """nearest-exact resize to (n-1)*s + k - 2*pad, then 3x3 s1 conv.  conv_blocks.py:72-131."""

    def __init__(self, in_ch, out_ch, kernel_size=2, stride=2, learnable_interpolation=True, use_res=False):
        super().__init__()
        self.learnable_interpolation = learnable_interpolation
        self._k, self._s = kernel_size, stride
        if learnable_interpolation:
            self.up_op = nn.Conv2d(in_ch, out_ch, 3, 1, 1, bias=True)
            if use_res:
                self.up_skip = nn.PixelShuffle(2)   # conv_blocks.py:114-115

    def calc_shape(self, spatial):
        pad = monai_padding(self._k, self._s)
        return tuple(int((n - 1) * self._s + self._k - 2 * pad) for n in spatial)

    def forward(self, x, emb=None):
        x_res = F.interpolate(x, size=self.calc_shape(x.shape[2:]), mode="nearest-exact")
        if not self.learnable_interpolation:
            return x_res
        y = self.up_op(x_res)
        if hasattr(self, "up_skip"):   # conv_blocks.py:125-126
            y = y + self.up_skip(x)
        return y


class DownBlock(nn.Module):
    """VAE encoder stage: BasicDown -> Attention -> ConvBlock.  conv_blocks.py:368-441."""

    def __init__(self, in_ch, out_ch, kernel_size, stride, downsample_kernel_size, norm_name, use_res_block=True,
                 learnable_interpolation=True, use_attention="none", emb_channels=None):
        super().__init__()
        enable_down = stride != 1
        down_out = out_ch if (learnable_interpolation and enable_down) else in_ch
        self.down_op = (
            BasicDown(in_ch, out_ch, downsample_kernel_size, stride, learnable_interpolation) if enable_down else nn.Identity()
        )
        self.attention = Attention(down_out, down_out, 8, down_out // 8, norm_name, emb_channels, 1, use_attention)
        Blk = UnetResBlock if use_res_block else UnetBasicBlock
        self.conv_block = Blk(down_out, out_ch, kernel_size, 1, norm_name, True, emb_channels)

    def forward(self, x, emb=None):
        x = self.down_op(x)
        x = self.attention(x, emb)
        return self.conv_block(x, emb)


class UpBlock(nn.Module):
    """VAE decoder stage: BasicUp -> (+skip) -> Attention -> ConvBlock.  conv_blocks.py:444-528."""

    def __init__(self, in_ch, out_ch, kernel_size, stride, upsample_kernel_size, norm_name, use_res_block=True,
                 learnable_interpolation=True, use_attention="none", emb_channels=None, skip_channels=0):
        super().__init__()
        enable_up = stride != 1
        skip_out = out_ch if (learnable_interpolation and enable_up) else in_ch + skip_channels
        self.learnable_interpolation = learnable_interpolation
        self.up_op = (
            BasicUp(in_ch, out_ch, upsample_kernel_size, stride, learnable_interpolation) if enable_up else nn.Identity()
        )
        self.attention = Attention(skip_out, skip_out, 8, skip_out // 8, norm_name, emb_channels, 1, use_attention)
        Blk = UnetResBlock if use_res_block else UnetBasicBlock
        self.conv_block = Blk(skip_out, out_ch, kernel_size, 1, norm_name, True, emb_channels)

    def forward(self, x_enc, x_skip=None, emb=None):
        x = self.up_op(x_enc)
        if x_skip is not None:
            x = x + x_skip if self.learnable_interpolation else torch.cat((x, x_skip), dim=1)
        x = self.attention(x, emb)
        return self.conv_block(x, emb)


# ----------------------------------------------------------------------------- attention
def compute_attention(q, k, v, num_heads, scale):
    """attention_blocks.py:35-43: softmax((q*s)^T (k*s)) v per head, [B,(h d),N] layout."""
    b, c, n = q.shape
    d = c // num_heads
    q = q.reshape(b * num_heads, d, n)
    k = k.reshape(b * num_heads, d, k.shape[-1])
    v = v.reshape(b * num_heads, d, v.shape[-1])
    attn = torch.einsum("b d i, b d j -> b i j", q * scale, k * scale).softmax(dim=-1)
    out = torch.einsum("b i j, b d j-> b d i", attn, v)
    return out.reshape(b, c, n)


class LinearTransformer(nn.Module):
    """GN -> q from x, k/v from embedding (or x_n) -> MHA -> zero-init 1x1 -> +x.  attention_blocks.py:128-195."""

    def __init__(self, in_ch, out_ch, num_heads, ch_per_head, norm_name, emb_dim=None):
        super().__init__()
        hid = num_heads * ch_per_head
        self.num_heads = num_heads
        self.scale = ch_per_head ** -0.25
        self.norm_x = _norm(norm_name, in_ch)
        emb_dim = in_ch if emb_dim is None else emb_dim
        self.to_q = nn.Conv1d(in_ch, hid, 1)
        self.to_k = nn.Conv1d(emb_dim, hid, 1)
        self.to_v = nn.Conv1d(emb_dim, hid, 1)
        self.to_out = nn.Sequential(zero_module(nn.Conv1d(hid, out_ch, 1)), nn.Identity())

    def forward(self, x, embedding=None):
        b, c, *spatial = x.shape
        x_n = self.norm_x(x)
        if embedding is None:
            embedding = x_n
        elif embedding.ndim == 2:
            embedding = embedding.reshape(*embedding.shape[:2], *[1] * (x.ndim - 2))
        x_n = x_n.reshape(b, c, -1)
        embedding = embedding.reshape(*embedding.shape[:2], -1)
        q, k, v = self.to_q(x_n), self.to_k(embedding), self.to_v(embedding)
        out = compute_attention(q, k, v, self.num_heads, self.scale)
        out = self.to_out(out)
        out = out.reshape(*out.shape[:2], *spatial)
        return x + out if x.shape == out.shape else out


class GEGLU(nn.Module):
    """LayerNorm over C, Linear C->2*out, x*gelu(gate).  attention_blocks.py:11-25."""

    def __init__(self, in_ch, out_ch):
        super().__init__()
        self.norm = nn.LayerNorm(in_ch)
        self.proj = nn.Linear(in_ch, out_ch * 2, bias=True)

    def forward(self, x):
        b, c, *spatial = x.shape
        x = x.reshape(b, c, -1).transpose(1, 2)
        x = self.norm(x)
        x, gate = self.proj(x).chunk(2, dim=-1)
        x = x * F.gelu(gate)
        return x.transpose(1, 2).reshape(b, -1, *spatial)


class BasicTransformerBlock(nn.Module):
    """self-attn, cross-attn (if emb), GEGLU FF + 1x1.  attention_blocks.py:200-231."""

    def __init__(self, in_ch, out_ch, num_heads, ch_per_head, norm_name, emb_dim=None):
        super().__init__()
        self.self_atn = LinearTransformer(in_ch, in_ch, num_heads, ch_per_head, norm_name, None)
        if emb_dim is not None:
            self.cros_atn = LinearTransformer(in_ch, in_ch, num_heads, ch_per_head, norm_name, emb_dim)
        self.proj_out = nn.Sequential(GEGLU(in_ch, in_ch * 4), nn.Identity(), nn.Conv2d(in_ch * 4, out_ch, 1, bias=True))

    def forward(self, x, embedding=None):
        x = self.self_atn(x)
        if embedding is not None:
            x = self.cros_atn(x, embedding=embedding)
        out = self.proj_out(x)
        return out + x if out.shape[1] == x.shape[1] else x


class SpatialTransformer(nn.Module):
    """GN -> 1x1 in -> blocks -> 1x1 out -> +x.  attention_blocks.py:233-288."""

    def __init__(self, in_ch, out_ch, num_heads, ch_per_head, norm_name, emb_dim=None, depth=1):
        super().__init__()
        self.norm = _norm(norm_name, in_ch)
        hid = num_heads * ch_per_head
        self.proj_in = nn.Conv2d(in_ch, hid, 1)
        self.transformer_blocks = nn.ModuleList(
            [BasicTransformerBlock(hid, hid, num_heads, ch_per_head, norm_name, emb_dim) for _ in range(depth)]
        )
        self.proj_out = nn.Conv2d(hid, out_ch, 1)

    def forward(self, x, embedding=None):
        h = self.proj_in(self.norm(x))
        for blk in self.transformer_blocks:
            h = blk(h, embedding=embedding)
        h = self.proj_out(h)
        return h + x if h.shape == x.shape else h


class Attention(nn.Module):
    """Switch 'none' | 'linear' | 'spatial'.  attention_blocks.py:291-335."""

    def __init__(self, in_ch, out_ch, num_heads=8, ch_per_head=32, norm_name=("GROUP", {"num_groups": 32, "affine": True}),
                 emb_dim=None, depth=1, attention_type="linear"):
        super().__init__()
        if attention_type == "spatial":
            self.attention = SpatialTransformer(in_ch, out_ch, num_heads, ch_per_head, norm_name, emb_dim, depth)
        elif attention_type == "linear":
            self.attention = LinearTransformer(in_ch, out_ch, num_heads, ch_per_head, norm_name, emb_dim)

    def forward(self, x, emb=None):
        return self.attention(x, emb) if hasattr(self, "attention") else x


# ----------------------------------------------------------------------------- embedders
class SinusoidalPosEmb(nn.Module):
    """time_embedder.py:7-28."""

    def __init__(self, emb_dim=16, downscale_freq_shift=1, max_period=10000, flip_sin_to_cos=False):
        super().__init__()
        self.emb_dim, self.downscale_freq_shift = emb_dim, downscale_freq_shift
        self.max_period, self.flip_sin_to_cos = max_period, flip_sin_to_cos

    def forward(self, x):
        half = self.emb_dim // 2
        e = math.log(self.max_period) / (half - self.downscale_freq_shift)
        e = torch.exp(-e * torch.arange(half, device=x.device))
        e = x[:, None] * e[None, :]
        e = torch.cat((e.sin(), e.cos()), dim=-1)
        if self.flip_sin_to_cos:
            e = torch.cat([e[:, half:], e[:, :half]], dim=-1)
        if self.emb_dim % 2 == 1:
            e = F.pad(e, (0, 1, 0, 0))
        return e


class LearnedSinusoidalPosEmb(nn.Module):
    """time_embedder.py:31-49: [x | sin(2 pi x w) | cos(2 pi x w)] with a learned frequency vector `weights` [emb_dim // 2]; the output has
    emb_dim + 1 features for an even emb_dim and emb_dim + 1 (zero-padded) for an odd one -- which is why the reference's own
    TimeEmbbeding(pos_embedder=LearnedSinusoidalPosEmb) raises in its first Linear (checked against the reference in gen_golden.py)."""

    def __init__(self, emb_dim):
        super().__init__()
        self.emb_dim = emb_dim
        self.weights = nn.Parameter(torch.randn(emb_dim // 2))

    def forward(self, x):
        x = x[:, None]
        freqs = x * self.weights[None, :] * 2 * math.pi
        f = torch.cat((freqs.sin(), freqs.cos()), dim=-1)
        f = torch.cat((x, f), dim=-1)
        if self.emb_dim % 2 == 1:
            f = F.pad(f, (0, 1, 0, 0))
        return f


class TimeEmbbeding(nn.Module):
    """sinusoid(emb_dim//4) -> Linear -> Swish -> Linear.  time_embedder.py:52-75.
    (Implements the intended `pos_emb_dim = emb_dim // 4` rule; the reference's mutable
    default dict can leak a previous instance's value -- SURVEY §8c trap.)"""

    def __init__(self, emb_dim=64, pos_embedder_kwargs=None):
        super().__init__()
        kw = dict(pos_embedder_kwargs or {})
        self.emb_dim = emb_dim
        self.pos_emb_dim = kw.get("emb_dim", emb_dim // 4)
        kw["emb_dim"] = self.pos_emb_dim
        self.pos_embedder = SinusoidalPosEmb(**kw)
        self.time_emb = nn.Sequential(
            self.pos_embedder, nn.Linear(self.pos_emb_dim, emb_dim), Swish(), nn.Linear(emb_dim, emb_dim)
        )

    def forward(self, time):
        return self.time_emb(time)


class LabelEmbedder(nn.Module):
    """cond_embedders.py:6-24."""

    def __init__(self, emb_dim=32, num_classes=2):
        super().__init__()
        self.emb_dim = emb_dim
        self.embedding = nn.Embedding(num_classes, emb_dim)

    def forward(self, condition):
        return self.embedding(condition)


class UnetOutBlock(nn.Module):
    """MONAI UnetOutBlock as used at unet2.py:213,217: 1x1 conv, keys `.conv.conv.*`."""

    def __init__(self, in_ch, out_ch):
        super().__init__()
        inner = nn.Sequential()
        inner.add_module("conv", nn.Conv2d(in_ch, out_ch, 1, 1, bias=True))
        self.conv = inner

    def forward(self, x):
        return self.conv(x)


# ----------------------------------------------------------------------------- UNet
class UNet(nn.Module):
    """unet2.py:15-269 (the exported UNet), spatial_dims=2 only."""

    def __init__(self, in_ch=1, out_ch=1, spatial_dims=2, hid_chs=(256, 256, 512, 1024), kernel_sizes=(3, 3, 3, 3),
                 strides=(1, 2, 2, 2), norm_name=("GROUP", {"num_groups": 32, "affine": True}),
                 time_embedder=TimeEmbbeding, time_embedder_kwargs=None, cond_embedder=None, cond_embedder_kwargs=None,
                 deep_supervision=True, use_res_block=True, estimate_variance=False, use_self_conditioning=False,
                 dropout=0.0, learnable_interpolation=True, use_attention="none", num_res_blocks=2):
        super().__init__()
        assert spatial_dims == 2
        hid_chs, kernel_sizes, strides = list(hid_chs), list(kernel_sizes), list(strides)
        use_attention = use_attention if isinstance(use_attention, list) else [use_attention] * len(strides)
        self.use_self_conditioning = use_self_conditioning
        self.depth = len(strides)
        self.num_res_blocks = num_res_blocks

        self.time_embedder = time_embedder(**dict(time_embedder_kwargs or {})) if time_embedder is not None else None
        temb = self.time_embedder.emb_dim if self.time_embedder is not None else None
        self.cond_embedder = cond_embedder(**dict(cond_embedder_kwargs or {})) if cond_embedder is not None else None

        Blk = UnetResBlock if use_res_block else UnetBasicBlock
        in_ch = in_ch * 2 if use_self_conditioning else in_ch
        self.in_conv = BasicBlock(in_ch, hid_chs[0], kernel_sizes[0], strides[0])

        def att(ch, lvl):
            return Attention(ch, ch, 8, ch // 8, norm_name, temb, 1, use_attention[lvl])

        in_blocks = []
        for i in range(1, self.depth):
            for k in range(num_res_blocks):
                in_blocks.append(SequentialEmb(
                    Blk(hid_chs[i - 1 if k == 0 else i], hid_chs[i], kernel_sizes[i], 1, norm_name, True, temb),
                    att(hid_chs[i], i)))
            if i < self.depth - 1:
                in_blocks.append(BasicDown(hid_chs[i], hid_chs[i], kernel_sizes[i], strides[i], learnable_interpolation))
        self.in_blocks = nn.ModuleList(in_blocks)

        self.middle_block = SequentialEmb(
            Blk(hid_chs[-1], hid_chs[-1], kernel_sizes[-1], 1, norm_name, True, temb),
            att(hid_chs[-1], -1),
            Blk(hid_chs[-1], hid_chs[-1], kernel_sizes[-1], 1, norm_name, True, temb))

        out_blocks = []
        for i in range(1, self.depth):
            for k in range(num_res_blocks + 1):
                oc = hid_chs[i - 1 if k == 0 else i]
                seq = [Blk(hid_chs[i] + oc, oc, kernel_sizes[i], 1, norm_name, True, temb), att(oc, i)]
                if i > 1 and k == 0:
                    seq.append(BasicUp(oc, oc, strides[i], strides[i], learnable_interpolation))
                out_blocks.append(SequentialEmb(*seq))
        self.out_blocks = nn.ModuleList(out_blocks)

        out_hor = out_ch * 2 if estimate_variance else out_ch
        self.outc = zero_module(UnetOutBlock(hid_chs[0], out_hor))
        if isinstance(deep_supervision, bool):
            deep_supervision = self.depth - 2 if deep_supervision else 0
        self.outc_ver = nn.ModuleList(
            [zero_module(UnetOutBlock(hid_chs[i] + hid_chs[i - 1], out_ch)) for i in range(2, deep_supervision + 2)])

    def forward(self, x_t, t=None, condition=None, self_cond=None):
        time_emb = None if t is None else self.time_embedder(t)
        cond_emb = None if (condition is None or self.cond_embedder is None) else self.cond_embedder(condition)
        emb = save_add(time_emb, cond_emb)
        if self.use_self_conditioning:  # reproduces the reference's `else x_t`
            self_cond = torch.zeros_like(x_t) if self_cond is None else x_t
            x_t = torch.cat([x_t, self_cond], dim=1)
        x = [self.in_conv(x_t)]
        for i in range(len(self.in_blocks)):
            x.append(self.in_blocks[i](x[i], emb))
        h = self.middle_block(x[-1], emb)
        y_ver = []
        for i in range(len(self.out_blocks), 0, -1):
            h = torch.cat([h, x.pop()], dim=1)
            depth, j = i // (self.num_res_blocks + 1), i % (self.num_res_blocks + 1) - 1
            if len(self.outc_ver) >= depth > 0 and j == 0:
                y_ver.append(self.outc_ver[depth - 1](h))
            h = self.out_blocks[i - 1](h, emb)
        return self.outc(h), y_ver[::-1]


# ----------------------------------------------------------------------------- VAE
class DiagonalGaussianDistribution(nn.Module):
    """latent_embedders.py:20-33.  `noise_fn(shape, device)` replaces torch.randn for injection."""

    def __init__(self):
        super().__init__()
        self.noise_fn: Optional[Callable] = None

    def forward(self, x):
        mean, logvar = torch.chunk(x, 2, dim=1)
        logvar = torch.clamp(logvar, -30.0, 20.0)
        std = torch.exp(0.5 * logvar)
        if self.noise_fn is None:
            sample = torch.randn(mean.shape, generator=None, device=x.device)
        else:
            sample = self.noise_fn(mean.shape, x.device)
        z = mean + std * sample
        var = torch.exp(logvar)
        kl = 0.5 * torch.sum(torch.pow(mean, 2) + var - 1.0 - logvar) / x.shape[0]
        return z, kl


class VAE(nn.Module):
    """latent_embedders.py:620-769 (encode/decode only; losses out of scope)."""

    def __init__(self, in_channels=3, out_channels=3, spatial_dims=2, emb_channels=4, hid_chs=(64, 128, 256, 512),
                 kernel_sizes=(3, 3, 3, 3), strides=(1, 2, 2, 2), norm_name=("GROUP", {"num_groups": 8, "affine": True}),
                 use_res_block=True, deep_supervision=False, learnable_interpolation=True, use_attention="none", **_ignored):
        super().__init__()
        assert spatial_dims == 2
        hid_chs, kernel_sizes, strides = list(hid_chs), list(kernel_sizes), list(strides)
        use_attention = use_attention if isinstance(use_attention, list) else [use_attention] * len(strides)
        self.depth = len(strides)
        Blk = UnetResBlock if use_res_block else UnetBasicBlock
        self.inc = Blk(in_channels, hid_chs[0], kernel_sizes[0], strides[0], norm_name, True, None)
        self.encoders = nn.ModuleList([
            DownBlock(hid_chs[i - 1], hid_chs[i], kernel_sizes[i], strides[i], kernel_sizes[i], norm_name, use_res_block,
                      learnable_interpolation, use_attention[i], None)
            for i in range(1, self.depth)])
        self.out_enc = nn.Sequential(BasicBlock(hid_chs[-1], 2 * emb_channels, 3), BasicBlock(2 * emb_channels, 2 * emb_channels, 1))
        self.quantizer = DiagonalGaussianDistribution()
        self.inc_dec = Blk(emb_channels, hid_chs[-1], 3, 1, norm_name, True, None)
        self.decoders = nn.ModuleList([
            UpBlock(hid_chs[i + 1], hid_chs[i], kernel_sizes[i + 1], strides[i + 1], strides[i + 1], norm_name, use_res_block,
                    learnable_interpolation, use_attention[i], None, 0)
            for i in range(self.depth - 1)])
        self.outc = BasicBlock(hid_chs[0], out_channels, 1, zero_conv=True)
        if isinstance(deep_supervision, bool):
            deep_supervision = self.depth - 1 if deep_supervision else 0
        self.outc_ver = nn.ModuleList([BasicBlock(hid_chs[i], out_channels, 1, zero_conv=True) for i in range(1, deep_supervision + 1)])

    def encode(self, x):
        h = self.inc(x)
        for enc in self.encoders:
            h = enc(h)
        z, _ = self.quantizer(self.out_enc(h))
        return z

    def decode(self, z):
        h = self.inc_dec(z)
        for i in range(len(self.decoders), 0, -1):
            h = self.decoders[i - 1](h)
        return self.outc(h)

    def forward(self, x_in):
        """latent_embedders.py:771-790: the reconstruction pass -> (out, deep-supervision outputs coarse-to-fine reversed, KL term)"""
        h = self.inc(x_in)
        for enc in self.encoders:
            h = enc(h)
        z_q, emb_loss = self.quantizer(self.out_enc(h))
        out_hor = []
        h = self.inc_dec(z_q)
        for i in range(len(self.decoders) - 1, -1, -1):
            if i < len(self.outc_ver):
                out_hor.append(self.outc_ver[i](h))
            h = self.decoders[i](h)
        return self.outc(h), out_hor[::-1], emb_loss


# ----------------------------------------------------------------------------- scheduler
class GaussianNoiseScheduler(nn.Module):
    """scheduler_base.py:7-46 + gaussian_scheduler.py:8-151.  `noise_fn(like)` replaces randn_like."""

    def __init__(self, timesteps=1000, T=None, schedule_strategy="cosine", beta_start=0.0001, beta_end=0.02, betas=None):
        super().__init__()
        self.timesteps = timesteps
        self.T = timesteps if T is None else T
        self.noise_fn: Optional[Callable] = None
        self.register_buffer("timesteps_array", torch.linspace(0, self.T - 1, self.timesteps, dtype=torch.long))
        self.schedule_strategy = schedule_strategy
        if betas is not None:
            betas = torch.as_tensor(betas, dtype=torch.float64)
        elif schedule_strategy == "linear":
            betas = torch.linspace(beta_start, beta_end, timesteps, dtype=torch.float64)
        elif schedule_strategy == "scaled_linear":
            betas = torch.linspace(beta_start ** 0.5, beta_end ** 0.5, timesteps, dtype=torch.float64) ** 2
        elif schedule_strategy == "cosine":
            s = 0.008
            x = torch.linspace(0, timesteps, timesteps + 1, dtype=torch.float64)
            ac = torch.cos(((x / timesteps) + s) / (1 + s) * torch.pi * 0.5) ** 2
            ac = ac / ac[0]
            betas = torch.clip(1 - (ac[1:] / ac[:-1]), 0, 0.999)
        else:
            raise NotImplementedError(schedule_strategy)
        alphas = 1 - betas
        ac = torch.cumprod(alphas, dim=0)
        ac_prev = F.pad(ac[:-1], (1, 0), value=1.0)
        reg = lambda n, v: self.register_buffer(n, v.to(torch.float32))
        reg("betas", betas)
        reg("alphas", alphas)
        reg("alphas_cumprod", ac)
        reg("alphas_cumprod_prev", ac_prev)
        reg("sqrt_alphas_cumprod", torch.sqrt(ac))
        reg("sqrt_one_minus_alphas_cumprod", torch.sqrt(1.0 - ac))
        reg("sqrt_recip_alphas_cumprod", torch.sqrt(1.0 / ac))
        reg("sqrt_recipm1_alphas_cumprod", torch.sqrt(1.0 / ac - 1))
        reg("posterior_mean_coef1", betas * torch.sqrt(ac_prev) / (1.0 - ac))
        reg("posterior_mean_coef2", (1.0 - ac_prev) * torch.sqrt(alphas) / (1.0 - ac))
        reg("posterior_variance", betas * (1.0 - ac_prev) / (1.0 - ac))

    @staticmethod
    def extract(x, t, ndim):
        return x.gather(0, t).reshape(-1, *((1,) * (ndim - 1)))

    def x_final(self, x):
        return torch.randn_like(x) if self.noise_fn is None else self.noise_fn(x)

    @staticmethod
    def _clip_x_0(x_0):
        return x_0.clamp(-1, 1)

    def sample(self, x_0):
        t = torch.randint(0, self.T, (x_0.shape[0],), dtype=torch.long, device=x_0.device)
        x_T = self.x_final(x_0)
        return self.estimate_x_t(x_0, t, x_T), x_T, t

    def estimate_x_t(self, x_0, t, x_T=None):
        x_T = self.x_final(x_0) if x_T is None else x_T

        def row(b):
            tb = t[b]
            if tb < 0:
                return x_0[b]
            if tb >= self.T:
                return x_T[b]
            return self.sqrt_alphas_cumprod[tb] * x_0[b] + self.sqrt_one_minus_alphas_cumprod[tb] * x_T[b]

        return torch.stack([row(b) for b in range(t.shape[0])])

    def estimate_x_t_prior_from_x_T(self, x_t, t, x_T, use_log=True, clip_x0=True, var_scale=0, cold_diffusion=False):
        x_0 = self.estimate_x_0(x_t, x_T, t, clip_x0)
        return self.estimate_x_t_prior_from_x_0(x_t, t, x_0, use_log, clip_x0, var_scale, cold_diffusion)

    def estimate_x_t_prior_from_x_0(self, x_t, t, x_0, use_log=True, clip_x0=True, var_scale=0, cold_diffusion=False):
        x_0 = self._clip_x_0(x_0) if clip_x0 else x_0
        if cold_diffusion:
            x_T_est = self.estimate_x_T(x_t, x_0, t)
            x_t_est = self.estimate_x_t(x_0, t, x_T=x_T_est)
            x_t_prior = self.estimate_x_t(x_0, t - 1, x_T=x_T_est)
            x_t_prior = x_t - (x_t_est - x_t_prior)
        else:
            mean = self.estimate_mean_t(x_t, x_0, t)
            variance = self.estimate_variance_t(t, x_t.ndim, use_log, var_scale)
            std = torch.exp(0.5 * variance) if use_log else torch.sqrt(variance)
            std[t == 0] = 0.0
            x_T = self.x_final(x_t)
            x_t_prior = mean + std * x_T
        return x_t_prior, x_0

    def estimate_mean_t(self, x_t, x_0, t):
        nd = x_t.ndim
        return self.extract(self.posterior_mean_coef1, t, nd) * x_0 + self.extract(self.posterior_mean_coef2, t, nd) * x_t

    def estimate_variance_t(self, t, ndim, log=True, var_scale=0, eps=1e-20):
        mn = self.extract(self.posterior_variance, t, ndim)
        mx = self.extract(self.betas, t, ndim)
        if log:
            mn = torch.log(mn.clamp(min=eps))
            mx = torch.log(mx.clamp(min=eps))
        return var_scale * mx + (1 - var_scale) * mn

    def estimate_x_0(self, x_t, x_T, t, clip_x0=True):
        nd = x_t.ndim
        x_0 = self.extract(self.sqrt_recip_alphas_cumprod, t, nd) * x_t - self.extract(self.sqrt_recipm1_alphas_cumprod, t, nd) * x_T
        return self._clip_x_0(x_0) if clip_x0 else x_0

    def estimate_x_T(self, x_t, x_0, t, clip_x0=True):
        nd = x_t.ndim
        x_0 = self._clip_x_0(x_0) if clip_x0 else x_0
        return (self.extract(self.sqrt_recip_alphas_cumprod, t, nd) * x_t - x_0) / self.extract(self.sqrt_recipm1_alphas_cumprod, t, nd)


# ----------------------------------------------------------------------------- pipeline
class _EMAHolder(nn.Module):
    """Holds `averaged_model` so keys read `ema_model.averaged_model.*` (train_utils.py:33-34)."""

    def __init__(self, model):
        super().__init__()
        import copy

        self.averaged_model = copy.deepcopy(model).eval()


class DiffusionPipeline(nn.Module):
    """Sampling half of diffusion_pipeline.py:20-332 (forward :232-275, denoise :278-310,
    sample :312-317, interpolate :320-332).  No streamlit/tqdm (Q16).  Takes *instances*."""

    def __init__(self, noise_scheduler: GaussianNoiseScheduler, noise_estimator: UNet, latent_embedder: Optional[VAE] = None,
                 estimator_objective="x_T", estimate_variance=False, use_self_conditioning=False, clip_x0=True, use_ema=False):
        super().__init__()
        self.noise_scheduler = noise_scheduler
        self.noise_estimator = noise_estimator
        self.latent_embedder = latent_embedder
        self.estimator_objective = estimator_objective
        self.estimate_variance = estimate_variance
        self.use_self_conditioning = use_self_conditioning
        self.clip_x0 = clip_x0
        self.use_ema = use_ema
        if use_ema:
            self.ema_model = _EMAHolder(noise_estimator)
        self.noise_fn: Optional[Callable] = None  # noise_fn(like) -> tensor

    @property
    def device(self):
        return next(self.parameters()).device

    def set_noise_fn(self, fn):
        """Inject one noise source for every randn_like on the path (Q3 draw order)."""
        self.noise_fn = fn
        self.noise_scheduler.noise_fn = fn

    def _randn_like(self, x):
        return torch.randn_like(x) if self.noise_fn is None else self.noise_fn(x)

    def forward(self, x_t, t, condition=None, self_cond=None, guidance_scale=1.0, cold_diffusion=False, un_cond=None):
        est = self.ema_model.averaged_model if self.use_ema else self.noise_estimator
        if (condition is not None) and (guidance_scale != 1.0):
            pred_uncond, _ = est(x_t, t, condition=un_cond, self_cond=self_cond)
            pred_cond, _ = est(x_t, t, condition=condition, self_cond=self_cond)
            pred = pred_uncond + guidance_scale * (pred_cond - pred_uncond)
            if self.estimate_variance:
                pred_uncond, pv_u = pred_uncond.chunk(2, dim=1)
                pred_cond, pv_c = pred_cond.chunk(2, dim=1)
                pred_var = pv_u + guidance_scale * (pv_c - pv_u)
        else:
            pred, _ = est(x_t, t, condition=condition, self_cond=self_cond)
            if self.estimate_variance:
                pred, pred_var = pred.chunk(2, dim=1)
        var_scale = pred_var / 2 + 0.5 if self.estimate_variance else 0
        sch = self.noise_scheduler
        if self.estimator_objective == "x_0":
            x_t_prior, x_0 = sch.estimate_x_t_prior_from_x_0(x_t, t, pred, clip_x0=self.clip_x0, var_scale=var_scale, cold_diffusion=cold_diffusion)
            x_T = sch.estimate_x_T(x_t, x_0=pred, t=t, clip_x0=self.clip_x0)
            self_cond = x_T
        elif self.estimator_objective == "x_T":
            x_t_prior, x_0 = sch.estimate_x_t_prior_from_x_T(x_t, t, pred, clip_x0=self.clip_x0, var_scale=var_scale, cold_diffusion=cold_diffusion)
            x_T = pred
            self_cond = x_0
        else:
            raise ValueError("Unknown Objective")
        return x_t_prior, x_0, x_T, self_cond

    @torch.no_grad()
    def denoise(self, x_t, steps=None, condition=None, use_ddim=True, trace=None, **kwargs):
        self_cond = None
        sch = self.noise_scheduler
        if use_ddim:
            steps = sch.timesteps if steps is None else steps
            timesteps_array = torch.linspace(0, sch.T - 1, steps, dtype=torch.long, device=x_t.device)
        else:
            timesteps_array = sch.timesteps_array[slice(0, steps)]
        for i, t in enumerate(reversed(timesteps_array)):
            x_t, x_0, x_T, self_cond = self(x_t, t.expand(x_t.shape[0]), condition, self_cond=self_cond, **kwargs)
            self_cond = self_cond if self.use_self_conditioning else None
            if use_ddim and (steps - i - 1 > 0):
                t_next = timesteps_array[steps - i - 2]
                alpha = sch.alphas_cumprod[t]
                alpha_next = sch.alphas_cumprod[t_next]
                sigma = kwargs.get("eta", 1) * ((1 - alpha / alpha_next) * (1 - alpha_next) / (1 - alpha)).sqrt()
                c = (1 - alpha_next - sigma ** 2).sqrt()
                noise = self._randn_like(x_t)
                x_t = x_0 * alpha_next.sqrt() + c * x_T + sigma * noise
            if trace is not None:
                trace.append((x_0.clone(), x_t.clone()))
        if self.latent_embedder is not None:
            x_t = self.latent_embedder.decode(x_t)
        return x_t

    @torch.no_grad()
    def sample(self, num_samples, img_size, condition=None, **kwargs):
        template = torch.zeros((num_samples, *img_size), device=self.device)
        x_T = self.noise_scheduler.x_final(template)
        return self.denoise(x_T, condition=condition, **kwargs)

    @torch.no_grad()
    def interpolate(self, img1, img2, i=None, condition=None, lam=0.5, **kwargs):
        assert img1.shape == img2.shape
        t = torch.full(img1.shape[:1], i, device=img1.device)
        a = self.noise_scheduler.estimate_x_t(img1, t=t)
        b = self.noise_scheduler.estimate_x_t(img2, t=t)
        img = (1 - lam) * a + lam * b
        return self.denoise(img, i, condition, **kwargs)


# ----------------------------------------------------------------------------- configs
def published_unet_kwargs(num_classes: Optional[int] = 2, in_ch: int = 8) -> dict:
    """scripts/train_diffusion.py:70-132."""
    kw = dict(in_ch=in_ch, out_ch=in_ch, spatial_dims=2, hid_chs=[256, 256, 512, 1024], kernel_sizes=[3, 3, 3, 3],
              strides=[1, 2, 2, 2], time_embedder=TimeEmbbeding, time_embedder_kwargs={"emb_dim": 1024},
              deep_supervision=False, use_res_block=True, use_attention="none")
    if num_classes is not None:
        kw.update(cond_embedder=LabelEmbedder, cond_embedder_kwargs={"emb_dim": 1024, "num_classes": num_classes})
    return kw


def published_vae_kwargs(emb_channels: int = 8) -> dict:
    """scripts/train_latent_embedder_2d.py:68-81."""
    return dict(in_channels=3, out_channels=3, emb_channels=emb_channels, spatial_dims=2, hid_chs=[64, 128, 256, 512],
                kernel_sizes=[3, 3, 3, 3], strides=[1, 2, 2, 2], deep_supervision=1, use_attention="none")


def published_scheduler_kwargs() -> dict:
    return dict(timesteps=1000, beta_start=0.002, beta_end=0.02, schedule_strategy="scaled_linear")


def tiny_unet_kwargs(num_classes: Optional[int] = 2, use_attention="none", hid=(32, 32, 64, 128), emb_dim=64, **extra) -> dict:
    kw = dict(in_ch=8, out_ch=8, spatial_dims=2, hid_chs=list(hid), kernel_sizes=[3, 3, 3, 3], strides=[1, 2, 2, 2],
              time_embedder=TimeEmbbeding, time_embedder_kwargs={"emb_dim": emb_dim}, deep_supervision=False,
              use_res_block=True, use_attention=use_attention)
    if num_classes is not None:
        kw.update(cond_embedder=LabelEmbedder, cond_embedder_kwargs={"emb_dim": emb_dim, "num_classes": num_classes})
    kw.update(extra)
    return kw


def tiny_vae_kwargs(hid=(32, 32, 64, 64), emb_channels=8) -> dict:
    return dict(in_channels=3, out_channels=3, emb_channels=emb_channels, spatial_dims=2, hid_chs=list(hid),
                kernel_sizes=[3, 3, 3, 3], strides=[1, 2, 2, 2], deep_supervision=1, use_attention="none")
