"""UNet noise estimator on the HIP kernels -- mirror of medical_diffusion/models/estimators/unet2.py:15-269
(the exported `UNet`) and of the embedders in models/embedders/{time_embedder,cond_embedders}.py.
Same constructor arguments, `forward(x_t, t, condition, self_cond) -> (y, y_ver)` and state-dict keys.
"""
from __future__ import annotations

import math
import os
from typing import Optional

import torch
import torch.nn as nn

from . import kernels as K
from . import lib as L
from . import blocks as BLK
from .blocks import (Attention, BasicBlock, BasicDown, BasicUp, Conv, SequentialEmb, UnetBasicBlock, UnetResBlock, _EmbBlock, zero_module)


class SinusoidalPosEmb(nn.Module):
    """time_embedder.py:7-28"""

    def __init__(self, emb_dim=16, downscale_freq_shift=1, max_period=10000, flip_sin_to_cos=False):
        super().__init__()
        self.emb_dim, self.downscale_freq_shift, self.max_period, self.flip_sin_to_cos = emb_dim, downscale_freq_shift, max_period, flip_sin_to_cos
        self._freqs = None

    def freqs(self, device):
        """exp(-ln(max_period)/(half - shift) * arange(half)) evaluated on the host with the reference's own ops
        (time_embedder.py:17-18), uploaded once: a 1-ulp difference here is amplified ~1000x by t."""
        if self._freqs is None or self._freqs.device != device:
            half = self.emb_dim // 2
            e = math.log(self.max_period) / (half - self.downscale_freq_shift)
            self._freqs = torch.exp(-e * torch.arange(half)).to(device)
        return self._freqs

    def forward(self, x):
        return K.sinusoidal(x, self.emb_dim, float(self.max_period), float(self.downscale_freq_shift), self.flip_sin_to_cos,
                            freqs=self.freqs(x.device))


class LearnedSinusoidalPosEmb(nn.Module):
    """time_embedder.py:31-49: [x | sin(2 pi x w) | cos(2 pi x w)] with the learned frequency vector `weights` [emb_dim // 2] (same key).
    The row has emb_dim + 1 features (even emb_dim; odd: emb_dim + 1 after the zero pad), so -- as in the reference, whose first Linear
    raises on it -- TimeEmbbeding cannot hold this embedder; it is a stand-alone module."""

    def __init__(self, emb_dim):
        super().__init__()
        self.emb_dim = emb_dim
        self.weights = nn.Parameter(torch.randn(emb_dim // 2))

    def forward(self, x):
        return K.learned_sinusoidal(x, self.weights, self.emb_dim)


class TimeEmbbeding(nn.Module):
    """time_embedder.py:52-75: sinusoid(emb_dim//4) -> Linear -> Swish -> Linear (keys time_emb.1.*, time_emb.3.*)."""

    def __init__(self, emb_dim=64, pos_embedder=SinusoidalPosEmb, pos_embedder_kwargs=None, act_name=("SWISH", {})):
        super().__init__()
        kw = dict(pos_embedder_kwargs or {})  # fresh dict: the reference mutates a shared default (SURVEY §8c trap)
        self.emb_dim = emb_dim
        self.pos_emb_dim = kw.get("emb_dim", emb_dim // 4)
        kw["emb_dim"] = self.pos_emb_dim
        self.pos_embedder = pos_embedder(**kw)
        self.time_emb = nn.Sequential(self.pos_embedder, nn.Linear(self.pos_emb_dim, emb_dim), nn.Identity(), nn.Linear(emb_dim, emb_dim))

    def forward(self, time):
        s = self.pos_embedder(time)
        l1, l2 = self.time_emb[1], self.time_emb[3]
        if s.shape[1] != l1.in_features:   # (LearnedSinusoidalPosEmb: emb_dim + 1 features -- the reference's nn.Linear raises the same way)
            raise RuntimeError(f"mat1 and mat2 shapes cannot be multiplied ({s.shape[0]}x{s.shape[1]} and {l1.in_features}x{l1.out_features})")
        h = K.linear(s, l1.weight, l1.bias, act_out=True)
        return K.linear(h, l2.weight, l2.bias)


class LabelEmbedder(nn.Module):
    """cond_embedders.py:6-24"""

    def __init__(self, emb_dim=32, num_classes=2, act_name=("SWISH", {})):
        super().__init__()
        self.emb_dim = emb_dim
        self.embedding = nn.Embedding(num_classes, emb_dim)

    def forward(self, condition):
        out = torch.zeros((condition.shape[0], self.emb_dim), dtype=torch.float32, device=condition.device)
        return K.embedding_add(self.embedding.weight, condition, out)


class UnetOutBlock(nn.Module):
    """MONAI UnetOutBlock as used at unet2.py:213,217: 1x1 conv, keys `.conv.conv.{weight,bias}`."""

    def __init__(self, spatial_dims, in_channels, out_channels, dropout=None):
        super().__init__()
        inner = nn.Sequential()
        inner.add_module("conv", Conv(in_channels, out_channels, 1, 1, 0))
        self.conv = inner

    def forward(self, x, out_layout=L.LAYOUT_NCHW, rows=None):
        return self.conv.conv(x, out_layout=out_layout, rows=rows)


# the conv blocks of a UNet write their outputs as fp16 pairs only wherever every reader takes pairs (UNet._pairs_only_outputs); 0: A/B switch
PAIRS_ONLY_BLOCK_OUTPUTS = os.environ.get("MEDFUSION_PAIRS_ONLY_OUTPUTS", "1") != "0"


class UNet(nn.Module):
    def __init__(self, in_ch=1, out_ch=1, spatial_dims=3, hid_chs=[256, 256, 512, 1024], kernel_sizes=[3, 3, 3, 3], strides=[1, 2, 2, 2],
                 act_name=("SWISH", {}), norm_name=("GROUP", {"num_groups": 32, "affine": True}), time_embedder=TimeEmbbeding,
                 time_embedder_kwargs={}, cond_embedder=None, cond_embedder_kwargs={}, deep_supervision=True, use_res_block=True,
                 estimate_variance=False, use_self_conditioning=False, dropout=0.0, learnable_interpolation=True, use_attention="none",
                 num_res_blocks=2):
        super().__init__()
        if spatial_dims != 2:
            raise NotImplementedError("the HIP sampling path is 2-D (published Medfusion models are spatial_dims=2)")
        if act_name[0].upper() != "SWISH":
            raise NotImplementedError("only the Swish activation is on the HIP path")
        use_attention = use_attention if isinstance(use_attention, list) else [use_attention] * len(strides)
        self.use_self_conditioning, self.use_res_block = use_self_conditioning, use_res_block
        self.depth, self.num_res_blocks = len(strides), num_res_blocks
        self.out_ch, self.estimate_variance = out_ch, estimate_variance

        self.time_embedder = time_embedder(**dict(time_embedder_kwargs)) if time_embedder is not None else None
        time_emb_dim = self.time_embedder.emb_dim if self.time_embedder is not None else None
        self.cond_embedder = cond_embedder(**dict(cond_embedder_kwargs)) if cond_embedder is not None else None

        ConvBlock = UnetResBlock if use_res_block else UnetBasicBlock
        in_ch = in_ch * 2 if use_self_conditioning else in_ch
        self.in_conv = BasicBlock(spatial_dims, in_ch, hid_chs[0], kernel_size=kernel_sizes[0], stride=strides[0])

        def att(ch, lvl):
            return Attention(spatial_dims, ch, ch, 8, ch // 8, norm_name, dropout, time_emb_dim, 1, use_attention[lvl])

        in_blocks = []
        for i in range(1, self.depth):
            for k in range(num_res_blocks):
                in_blocks.append(SequentialEmb(
                    ConvBlock(spatial_dims, hid_chs[i - 1 if k == 0 else i], hid_chs[i], kernel_sizes[i], 1, norm_name, act_name, dropout, time_emb_dim),
                    att(hid_chs[i], i)))
            if i < self.depth - 1:
                in_blocks.append(BasicDown(spatial_dims, hid_chs[i], hid_chs[i], kernel_sizes[i], strides[i], learnable_interpolation))
        self.in_blocks = nn.ModuleList(in_blocks)

        self.middle_block = SequentialEmb(
            ConvBlock(spatial_dims, hid_chs[-1], hid_chs[-1], kernel_sizes[-1], 1, norm_name, act_name, dropout, time_emb_dim),
            att(hid_chs[-1], -1),
            ConvBlock(spatial_dims, hid_chs[-1], hid_chs[-1], kernel_sizes[-1], 1, norm_name, act_name, dropout, time_emb_dim))

        out_blocks = []
        for i in range(1, self.depth):
            for k in range(num_res_blocks + 1):
                oc = hid_chs[i - 1 if k == 0 else i]
                seq = [ConvBlock(spatial_dims, hid_chs[i] + oc, oc, kernel_sizes[i], 1, norm_name, act_name, dropout, time_emb_dim), att(oc, i)]
                if i > 1 and k == 0:
                    seq.append(BasicUp(spatial_dims, oc, oc, strides[i], strides[i], learnable_interpolation))
                out_blocks.append(SequentialEmb(*seq))
        self.out_blocks = nn.ModuleList(out_blocks)

        out_ch_hor = out_ch * 2 if estimate_variance else out_ch
        self.outc = zero_module(UnetOutBlock(spatial_dims, hid_chs[0], out_ch_hor, dropout=None))
        if isinstance(deep_supervision, bool):
            deep_supervision = self.depth - 2 if deep_supervision else 0
        self.outc_ver = nn.ModuleList([
            zero_module(UnetOutBlock(spatial_dims, hid_chs[i] + hid_chs[i - 1], out_ch, dropout=None)) for i in range(2, deep_supervision + 2)])

        # all local embedders (Swish -> Linear(E, Cout)) batched into ONE GEMM per forward
        self._emb_blocks = [m for m in self.modules() if isinstance(m, _EmbBlock) and hasattr(m, "local_embedder")]
        self._emb_cache_key = None
        self._emb_w = self._emb_b = None
        self._emb_off = {}
        self._po_cache = {}

    # ------------------------------------------------------------------ embeddings
    def _packed_local_embedders(self):
        lins = [m.local_embedder[1] for m in self._emb_blocks]
        key = tuple((l.weight.data_ptr(), l.weight._version, l.bias._version) for l in lins)
        if key != self._emb_cache_key:
            with torch.no_grad():
                self._emb_w = torch.cat([l.weight for l in lins], dim=0).contiguous()  # load-time packing (plumbing)
                self._emb_b = torch.cat([l.bias for l in lins], dim=0).contiguous()
            off = 0
            self._emb_off = {}
            for m, l in zip(self._emb_blocks, lins):
                self._emb_off[id(m)] = (off, l.weight.shape[0])
                off += l.weight.shape[0]
            self._emb_cache_key = key
        return self._emb_w, self._emb_b

    def embed(self, t, condition, emb_override=None, emb_cache=None):
        """Global embedding [B,E] and the lookup giving every module its embedding (unet2.py:229-241).
        emb_cache = (emb [B,E], local_all [B, sum Cout]) from `step_embeddings`: nothing is computed here."""
        if emb_cache is not None:
            emb, local_all = emb_cache
        else:
            time_emb = None if t is None else self.time_embedder(t)
            if emb_override is not None:
                emb = emb_override
            elif condition is None or self.cond_embedder is None:
                emb = time_emb
            elif time_emb is None:
                emb = self.cond_embedder(condition)
            else:  # save_add(time_emb, cond_emb): the lookup accumulates into the time embedding
                emb = K.embedding_add(self.cond_embedder.embedding.weight, condition, time_emb)
            if emb is None or not self._emb_blocks:
                return emb, (lambda m: emb if isinstance(m, Attention) else None)
            w, b = self._packed_local_embedders()
            local_all = K.linear(emb, w, b, act_in=True)  # [B, sum Cout]

        rowmax = None
        if BLK.f16x2_mode():  # operand bound of everything an embedding row is added to (fp16-pair scaling, kernels.gn_apply)
            rowmax = getattr(local_all, "_mf_bound", None)
            if rowmax is None:
                rowmax = K.maxabs_rows(local_all)
                local_all._mf_bound = rowmax

        def lookup(m):
            if isinstance(m, Attention):
                return emb
            o = self._emb_off.get(id(m))
            if o is None:
                return None
            v = local_all[:, o[0]:o[0] + o[1]]
            if rowmax is not None:
                v._mf_bound = rowmax
            return v

        return emb, lookup

    # The sampling loop knows all of its timesteps before it starts, every row of a step shares the timestep, and a row's
    # embedding depends on (timestep, class) only.  So the whole embedding path of unet2.py:229-241 + the local embedders of
    # conv_blocks.py:340-353 -- sinusoid, time MLP, label lookup + save_add, 17 Linear(Swish(.)) -- is evaluated ONCE per sample()
    # for the distinct (step, class) pairs and gathered per step.  Same kernels on the same values row by row (every kernel on this
    # path treats rows independently), so the result is bit-identical to evaluating it inside the loop (tests: hipgraph == eager).
    def can_precompute_embeddings(self) -> bool:
        return (self.time_embedder is not None and bool(self._emb_blocks)
                and (self.cond_embedder is None or isinstance(self.cond_embedder, LabelEmbedder)))

    @torch.no_grad()
    def precompute_embeddings(self, t_steps: torch.Tensor, classes=None):
        """t_steps [S] (the loop's timesteps in loop order) -> table: column 0 = no condition, column 1 + k = class classes[k].
        `classes`: the labels that actually occur (condition and un_cond of the loop); None = every class of the embedder.  The table
        costs S x (1 + len(classes)) rows, so a 1000-class embedder does not build 1001 columns for a batch that uses three."""
        S = t_steps.shape[0]
        time_emb = self.time_embedder(t_steps.to(torch.float32).contiguous())          # [S, E]
        cols = [time_emb]
        lut = None
        if self.cond_embedder is not None:
            tab = self.cond_embedder.embedding.weight
            classes = list(range(tab.shape[0])) if classes is None else sorted(set(int(c) for c in classes))
            lut = torch.zeros((tab.shape[0],), dtype=torch.long, device=t_steps.device)
            for k, c in enumerate(classes):
                lab = torch.full((S,), c, dtype=torch.long, device=t_steps.device)
                cols.append(K.embedding_add(tab, lab, time_emb.clone()))
                lut[c] = 1 + k
        emb = torch.stack(cols, dim=1).contiguous()                                     # [S, NCOL, E]  (plumbing)
        w, b = self._packed_local_embedders()
        local = K.linear(emb.view(S * len(cols), -1), w, b, act_in=True).view(S, len(cols), -1).contiguous()
        table = {"emb": emb, "local": local, "need_emb": any(isinstance(m, Attention) for m in self.modules()), "lut": lut}
        if BLK.f16x2_mode():  # bound of every embedding row, once (operand scaling of the fp16-pair mode)
            table["local_bound"] = K.maxabs_rows(local.view(S * len(cols), -1)).view(S, len(cols), 1).contiguous()
        return table

    @staticmethod
    def embedding_columns(condition, B, device, table=None):
        """row -> column of the precomputed table: 0 without a condition, else the column of its label"""
        if condition is None:
            return torch.zeros((B,), dtype=torch.long, device=device)
        lab = condition.to(device=device, dtype=torch.long).reshape(-1)
        if table is not None and table.get("lut") is not None:
            return table["lut"].index_select(0, lab)
        return lab + 1

    @staticmethod
    def step_embeddings(table, i, cols: torch.Tensor):
        """(emb [B,E] | None, local_all [B, sum Cout]) of loop iteration i for rows with table columns `cols`.
        i: a host int, or the device int32 step counter of a captured graph (mf_gather_step_rows_f32 reads it on the device)."""
        names = (["emb"] if table["need_emb"] else []) + ["local"] + (["local_bound"] if "local_bound" in table else [])
        got = dict(zip(names, K.gather_step_rows_multi([table[k] for k in names], i, cols)))    # ONE launch for the two or three gathers
        local = got["local"]
        if "local_bound" in got:
            local._mf_bound = got["local_bound"].view(-1)
        return got.get("emb"), local

    @torch.no_grad()
    def forward_cfg_pair(self, x_t, t, condition, un_cond, emb_cache=None):
        """Classifier-free-guidance pair in one pass: returns y [2B,...] with rows [0,B) = forward(x_t, t, un_cond) and
        rows [B,2B) = forward(x_t, t, condition).  Per-row arithmetic is the same as two separate calls."""
        B = x_t.shape[0]
        x2 = torch.empty((2 * B, *x_t.shape[1:]), dtype=x_t.dtype, device=x_t.device)   # plumbing: 2 x 8192*B floats, through the library
        K.rows_axpby(x_t, out=x2[:B])                  # (launches of the library only inside the loop body: the command-list loop
        K.rows_axpby(x_t, out=x2[B:])                  #  re-issues exactly those -- pipeline.py)
        if emb_cache is not None:
            h, _ = self.features(x2, None, None, None, emb_cache=emb_cache)
            return self.outc(h)
        t2 = torch.cat([t, t], dim=0)
        time_emb = self.time_embedder(t2)              # [2B, E]
        if self.cond_embedder is not None:
            tab = self.cond_embedder.embedding.weight
            if un_cond is not None:
                K.embedding_add(tab, un_cond, time_emb[:B])
            K.embedding_add(tab, condition, time_emb[B:])
        h, _ = self.features(x2, None, None, None, emb_override=time_emb)
        return self.outc(h)

    # ------------------------------------------------------------------ forward
    @torch.no_grad()
    def features(self, x_t, t=None, condition=None, self_cond=None, emb_override=None, emb_cache=None):
        """Everything of unet2.py:222-264 up to (not including) the 1x1 out convolution: (h NHWC, y_ver)."""
        if not x_t.is_cuda:
            raise RuntimeError("medfusion_amd.UNet runs on a ROCm device only (no CPU fallback)")
        x_t = x_t.contiguous()
        _, lookup = self.embed(t, condition, emb_override, emb_cache)
        if self.use_self_conditioning:
            # reference concatenates zeros if self_cond is None else x_t ITSELF (unet2.py:245)
            a = K.nchw_to_nhwc(x_t)
            b = torch.zeros_like(a) if self_cond is None else a
            h0 = self.in_conv((a, b))
        else:
            h0 = self.in_conv(x_t, in_layout=L.LAYOUT_NCHW)
        # Every conv block's output is read by fp16-pair convolutions, residual adds and skip concats only -- except the last one's (outc reads
        # fp32): where that holds for the whole network (round 5; checked once per input shape, below) the blocks do not write their fp32 form
        # at all (12 instead of 16 bytes per element behind every block: ~130 MB per evaluation at cfg2)
        po = self._pairs_only_outputs(h0.shape)
        x = [h0]
        for blk in self.in_blocks:
            x.append(blk(x[-1], lookup, out_fp32=not po) if isinstance(blk, SequentialEmb) else blk(x[-1]))
        h = self.middle_block(x[-1], lookup, out_fp32=not po)
        y_ver = []
        for i in range(len(self.out_blocks), 0, -1):
            hs = (h, x.pop())  # torch.cat([h, skip], 1) fused into the consumers
            depth, j = i // (self.num_res_blocks + 1), i % (self.num_res_blocks + 1) - 1
            if len(self.outc_ver) >= depth > 0 and j == 0:
                y_ver.append(self.outc_ver[depth - 1](hs))
            h = self.out_blocks[i - 1](hs, lookup, out_fp32=(i == 1) or not po)
        return h, y_ver[::-1]

    def _pairs_only_outputs(self, shape) -> bool:
        """may the conv blocks skip the fp32 form of their outputs for an in_conv output of this shape?  Only when every reader is a fp16-pair
        convolution for ITS shape: the default arithmetic, ResBlocks, no attention, no deep-supervision heads (they read fp32), learnable
        down / up convolutions; every block asked with the shape and the skip width it will see.  Cached per shape."""
        key = (tuple(shape), BLK.CONV_PRECISION, BLK.PAIRS_ONLY_BETWEEN_BLOCKS, PAIRS_ONLY_BLOCK_OUTPUTS)
        ok = self._po_cache.get(key)
        if ok is not None:
            return ok
        ok = (PAIRS_ONLY_BLOCK_OUTPUTS and BLK.PAIRS_ONLY_BETWEEN_BLOCKS and BLK.f16x2_mode() and self.use_res_block and len(self.outc_ver) == 0
              and not any(hasattr(m, "attention") for m in self.modules() if isinstance(m, Attention))
              and all(m.learnable and not m.use_res for m in self.modules() if isinstance(m, (BasicDown, BasicUp))))
        if ok:
            n, h, w, _ = shape
            hw = [(h, w)]
            widths = [shape[3]]                       # channels of x[0], x[1], ... (the skips)
            for blk in self.in_blocks:
                if isinstance(blk, SequentialEmb):
                    rb = blk[0].block_seq[0]
                    ok = ok and rb.reads_pairs_only(n, *hw[-1]) if len(widths) > 1 else ok     # (x[0] = in_conv's output has its fp32 form)
                    widths.append(blk[0].out_channels)
                    hw.append(hw[-1])
                else:
                    c = blk.down_op
                    ok = ok and K.conv_f16x2_ok(K.make_conv_desc(n, *hw[-1], c.in_ch, 0, c.out_ch, c.k, c.stride, c.pad, 0, precision=BLK.CONV_PRECISION))
                    widths.append(c.out_ch)
                    hw.append(((hw[-1][0] + 2 * c.pad - c.k) // c.stride + 1, (hw[-1][1] + 2 * c.pad - c.k) // c.stride + 1))
            cur = hw[-1]
            ok = ok and self.middle_block[0].block_seq[0].reads_pairs_only(n, *cur) and self.middle_block[2].block_seq[0].reads_pairs_only(n, *cur)
            for i in range(len(self.out_blocks), 0, -1):
                seq = self.out_blocks[i - 1]
                ok = ok and hw.pop() == cur and seq[0].block_seq[0].reads_pairs_only(n, *cur, c2=widths.pop())
                for m in list(seq)[1:]:
                    if isinstance(m, BasicUp):
                        c = m.up_op
                        ok = ok and K.conv_f16x2_ok(K.make_conv_desc(n, *cur, c.in_ch, 0, c.out_ch, 3, 1, 1, 2, precision=BLK.CONV_PRECISION))
                        cur = (2 * cur[0], 2 * cur[1])
        self._po_cache[key] = bool(ok)
        return bool(ok)

    @torch.no_grad()
    def forward(self, x_t, t=None, condition=None, self_cond=None, emb_cache=None):
        """x_t [B,C,H,W] NCHW on the GPU; t [B] (long or float); condition [B] long | None.
        Returns (y NCHW, y_ver list) like unet2.py:222-269.  emb_cache: see `step_embeddings` (sampling loop only)."""
        h, y_ver = self.features(x_t, t, condition, self_cond, emb_cache=emb_cache)
        return self.outc(h), y_ver

    @torch.no_grad()
    def forward_split(self, x_t, t=None, condition=None, self_cond=None, emb_cache=None):
        """estimate_variance=True: (pred, pred_var) == y.chunk(2, dim=1) of diffusion_pipeline.py:252 as two
        contiguous tensors (the two halves of `outc` run as two 1x1 convolutions over the same features)."""
        assert self.estimate_variance
        h, _ = self.features(x_t, t, condition, self_cond, emb_cache=emb_cache)
        c = self.out_ch
        return self.outc(h, rows=slice(0, c)), self.outc(h, rows=slice(c, 2 * c))
