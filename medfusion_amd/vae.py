"""VAE latent embedder on the HIP kernels -- mirror of medical_diffusion/models/embedders/latent_embedders.py
`VAE` (:620-769: ctor, encode :756-762, decode :764-769) and `DiagonalGaussianDistribution` (:20-33).
Training losses / perceiver / optimiser arguments are accepted and ignored (out of scope).
"""
from __future__ import annotations

from typing import Optional

import torch
import torch.nn as nn

from . import kernels as K
from . import lib as L
from .blocks import BasicBlock, DownBlock, UnetBasicBlock, UnetResBlock, UpBlock
from .noise import NoiseSource, default_noise


class DiagonalGaussianDistribution(nn.Module):
    """z = mean + exp(0.5*clamp(logvar,-30,20)) * N(0,1); KL is computed by the reference and discarded by encode."""

    def forward(self, moments_nchw: torch.Tensor, noise: torch.Tensor):
        return K.diag_gaussian_sample(moments_nchw, noise), None


class VAE(nn.Module):
    def __init__(self, in_channels=3, out_channels=3, spatial_dims=2, emb_channels=4, hid_chs=[64, 128, 256, 512], kernel_sizes=[3, 3, 3, 3],
                 strides=[1, 2, 2, 2], norm_name=("GROUP", {"num_groups": 8, "affine": True}), act_name=("Swish", {}), dropout=None,
                 use_res_block=True, deep_supervision=False, learnable_interpolation=True, use_attention="none", **_training_only):
        super().__init__()
        if spatial_dims != 2:
            raise NotImplementedError("the HIP sampling path is 2-D")
        use_attention = use_attention if isinstance(use_attention, list) else [use_attention] * len(strides)
        self.depth = len(strides)
        self.emb_channels = emb_channels
        self.out_channels = out_channels
        self.scale = 1                       # spatial factor between the latent and the image
        for st in strides:
            self.scale *= int(st)
        ConvBlock = UnetResBlock if use_res_block else UnetBasicBlock
        self.inc = ConvBlock(spatial_dims, in_channels, hid_chs[0], kernel_size=kernel_sizes[0], stride=strides[0], act_name=act_name,
                             norm_name=norm_name, emb_channels=None)
        self.encoders = nn.ModuleList([
            DownBlock(spatial_dims, hid_chs[i - 1], hid_chs[i], kernel_sizes[i], strides[i], kernel_sizes[i], norm_name, act_name, dropout,
                      use_res_block, learnable_interpolation, use_attention[i], None)
            for i in range(1, self.depth)])
        self.out_enc = nn.Sequential(BasicBlock(spatial_dims, hid_chs[-1], 2 * emb_channels, 3), BasicBlock(spatial_dims, 2 * emb_channels, 2 * emb_channels, 1))
        self.quantizer = DiagonalGaussianDistribution()
        self.inc_dec = ConvBlock(spatial_dims, emb_channels, hid_chs[-1], 3, act_name=act_name, norm_name=norm_name)
        self.decoders = nn.ModuleList([
            UpBlock(spatial_dims, hid_chs[i + 1], hid_chs[i], kernel_sizes[i + 1], strides[i + 1], strides[i + 1], norm_name, act_name, dropout,
                    use_res_block, learnable_interpolation, use_attention[i], None, 0)
            for i in range(self.depth - 1)])
        self.outc = BasicBlock(spatial_dims, hid_chs[0], out_channels, 1, zero_conv=True)
        if isinstance(deep_supervision, bool):
            deep_supervision = self.depth - 1 if deep_supervision else 0
        self.outc_ver = nn.ModuleList([BasicBlock(spatial_dims, hid_chs[i], out_channels, 1, zero_conv=True) for i in range(1, deep_supervision + 1)])

    @torch.no_grad()
    def encode(self, x: torch.Tensor, noise: Optional[NoiseSource] = None) -> torch.Tensor:
        """x [B,3,H,W] NCHW -> z [B,emb,H/8,W/8] NCHW (stochastic: one N(0,1) draw of z's shape)."""
        if not x.is_cuda:
            raise RuntimeError("medfusion_amd.VAE runs on a ROCm device only (no CPU fallback)")
        K.SyncWords.reset(x.device)

        moments = K.with_fused_fallback(x.device, lambda: self._encode_moments(x))
        n, c2, hh, ww = moments.shape
        src = noise if noise is not None else default_noise()
        src.begin(n, x.device)
        eps = src.draw((n, c2 // 2, hh, ww))
        z, _ = self.quantizer(moments, eps)
        return z

    @torch.no_grad()
    def decode(self, z: torch.Tensor) -> torch.Tensor:
        """z [B,emb,h,w] NCHW -> x [B,3,8h,8w] NCHW."""
        if not z.is_cuda:
            raise RuntimeError("medfusion_amd.VAE runs on a ROCm device only (no CPU fallback)")
        if z.shape[0] == 0:   # an empty shard of a multi-GPU batch (more ranks than samples): nothing to launch
            return z.new_empty((0, self.out_channels, z.shape[2] * self.scale, z.shape[3] * self.scale))
        K.SyncWords.reset(z.device)

        def run():
            h = self.inc_dec(z.contiguous(), None, in_layout=L.LAYOUT_NCHW)
            for i in range(len(self.decoders), 0, -1):
                h = self.decoders[i - 1](h)
            return self.outc(h, out_layout=L.LAYOUT_NCHW)

        return K.with_fused_fallback(z.device, run)

    def _encode_moments(self, x):
        h = self.inc(x.contiguous(), None, in_layout=L.LAYOUT_NCHW)
        for enc in self.encoders:
            h = enc(h)
        h = self.out_enc[0](h)
        return self.out_enc[1](h, out_layout=L.LAYOUT_NCHW)

    @torch.no_grad()
    def forward(self, x_in: torch.Tensor, noise: Optional[NoiseSource] = None):
        """latent_embedders.py:771-790 -- the reconstruction pass of the evaluation harness: (out [B,3,H,W], the deep-supervision outputs of
        the coarser decoder levels (finest first, like the reference's `out_hor[::-1]`), the KL term of the quantizer).  Inference only:
        the losses built on these (`_step`, :803-840) are training code and out of scope."""
        if not x_in.is_cuda:
            raise RuntimeError("medfusion_amd.VAE runs on a ROCm device only (no CPU fallback)")
        K.SyncWords.reset(x_in.device)
        src = noise if noise is not None else default_noise()

        drawn = []   # the quantizer noise is drawn ONCE: a re-run of the pass (with_fused_fallback) must see the same draw -- a host source cannot be rewound

        def run():
            moments = self._encode_moments(x_in)
            n, c2, hh, ww = moments.shape
            if not drawn:
                src.begin(n, x_in.device)
                drawn.append(src.draw((n, c2 // 2, hh, ww)))
            z_q, _ = self.quantizer(moments, drawn[0])
            emb_loss = K.diag_gaussian_kl(moments)
            out_hor = []
            h = self.inc_dec(z_q.contiguous(), None, in_layout=L.LAYOUT_NCHW)
            for i in range(len(self.decoders) - 1, -1, -1):
                if i < len(self.outc_ver):
                    out_hor.append(self.outc_ver[i](h, out_layout=L.LAYOUT_NCHW))
                h = self.decoders[i](h)
            return self.outc(h, out_layout=L.LAYOUT_NCHW), out_hor[::-1], emb_loss

        return K.with_fused_fallback(x_in.device, run)
