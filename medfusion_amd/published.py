"""The published Medfusion configuration and a seeded stand-in for its trained weights.

* Architecture of the released chest-X-ray model: UNet `scripts/train_diffusion.py:70-132`, VAE `scripts/train_latent_embedder_2d.py:68-81`,
  scheduler `scripts/train_diffusion.py:115-123`.  Citations are relative to the reference repo.
* No checkpoint, dataset or network exists where this runs, so `bench.py`, `scripts/sample.py --synthetic` and
  `scripts/sample_dataset.py --synthetic` fill every parameter from a hash of its state-dict KEY (`seeded_fill`): reproducible anywhere
  without files, non-degenerate (the reference zero-initialises 106 tensors), and identical to what the parity tests give
  the CPU oracle (tests/test_host_logic_cpu.py::test_seeded_fill_matches_the_test_fill pins the two against each other).
"""
from __future__ import annotations

import zlib
from typing import Optional

import numpy as np
import torch


def published_unet_kwargs(num_classes: Optional[int] = 2, in_ch: int = 8) -> dict:
    from .unet import LabelEmbedder, TimeEmbbeding

    kw = dict(in_ch=in_ch, out_ch=in_ch, spatial_dims=2, hid_chs=[256, 256, 512, 1024], kernel_sizes=[3, 3, 3, 3], strides=[1, 2, 2, 2],
              time_embedder=TimeEmbbeding, time_embedder_kwargs={"emb_dim": 1024}, deep_supervision=False, use_res_block=True, use_attention="none")
    if num_classes is not None:
        kw.update(cond_embedder=LabelEmbedder, cond_embedder_kwargs={"emb_dim": 1024, "num_classes": num_classes})
    return kw


def published_vae_kwargs(emb_channels: int = 8) -> dict:
    return dict(in_channels=3, out_channels=3, emb_channels=emb_channels, spatial_dims=2, hid_chs=[64, 128, 256, 512], kernel_sizes=[3, 3, 3, 3],
                strides=[1, 2, 2, 2], deep_supervision=1, use_attention="none")


def published_scheduler_kwargs() -> dict:
    return dict(timesteps=1000, beta_start=0.002, beta_end=0.02, schedule_strategy="scaled_linear")


# ----------------------------------------------------------------------------- seeded weights
def _mix(h: np.ndarray) -> np.ndarray:
    """murmur3's 32-bit finaliser"""
    h = h ^ (h >> np.uint32(16))
    h = h * np.uint32(0x85EBCA6B)
    h = h ^ (h >> np.uint32(13))
    h = h * np.uint32(0xC2B2AE35)
    return h ^ (h >> np.uint32(16))


def hashed_uniform(name: str, n: int, scale: float = 1.0, shift: float = 0.0) -> np.ndarray:
    """u_i in (-1, 1), i < n: the top 24 bits of mix(mix(i * 0x9E3779B1 + crc32(name))), centred; then shift + scale * u."""
    out = np.empty(n, dtype=np.float32)
    key = np.uint32(zlib.crc32(name.encode()))
    step = 1 << 22
    with np.errstate(over="ignore"):
        for lo in range(0, n, step):
            i = np.arange(lo, min(n, lo + step), dtype=np.uint32)
            h = _mix(_mix(i * np.uint32(0x9E3779B1) + key))
            f = ((h >> np.uint32(8)).astype(np.float32) + np.float32(0.5)) * np.float32(2.0 ** -23) - np.float32(1.0)
            if scale != 1.0:
                f *= np.float32(scale)
            if shift != 0.0:
                f += np.float32(shift)
            out[lo:lo + len(i)] = f
    return out


def _fill_rule(key: str, shape) -> tuple:
    """(scale, shift) by the role the key name gives a tensor: conv/linear weight ~ U * sqrt(3 / fan_in), bias ~ 0.1 U,
    norm gamma ~ 1 + 0.2 U, label-embedding table ~ U."""
    leaf = key.rsplit(".", 1)[-1]
    if "embedding" in key and leaf == "weight" and len(shape) == 2 and ".local_embedder" not in key and "time_emb" not in key:
        return 1.0, 0.0
    if leaf == "weight":
        if len(shape) == 1:
            return 0.2, 1.0
        fan_in = int(np.prod(shape[1:]))
        return float(np.float32(np.sqrt(3.0 / max(fan_in, 1)))), 0.0
    if leaf == "bias":
        return 0.1, 0.0
    raise ValueError(f"seeded_fill: unexpected parameter {key}")


@torch.no_grad()
def seeded_fill(module: torch.nn.Module, prefix: str = "") -> torch.nn.Module:
    """Overwrite every parameter of `module` from its (prefixed) key name; buffers are left alone."""
    for key, p in module.named_parameters():
        sc, sh = _fill_rule(key, tuple(p.shape))
        p.copy_(torch.from_numpy(hashed_uniform(prefix + key, p.numel(), sc, sh)).view(p.shape))
    return module


def build_published_pipeline(device=None, num_classes: Optional[int] = 2, emb_channels: int = 8, unet_prefix: str = "published.unet.",
                             vae_prefix: str = "published.vae."):
    """DiffusionPipeline of the published architecture with seeded weights (objective x_T, clip_x0=False like the released run)."""
    from .pipeline import DiffusionPipeline
    from .scheduler import GaussianNoiseScheduler
    from .unet import UNet
    from .utils import no_init
    from .vae import VAE

    with no_init():  # every parameter is overwritten below
        pipe = DiffusionPipeline(GaussianNoiseScheduler, UNet, None, published_scheduler_kwargs(), published_unet_kwargs(num_classes, emb_channels),
                                 estimator_objective="x_T", clip_x0=False)
        pipe.latent_embedder = VAE(**published_vae_kwargs(emb_channels))
    seeded_fill(pipe.noise_estimator, unet_prefix)
    seeded_fill(pipe.latent_embedder, vae_prefix)
    if device is not None:
        pipe = pipe.to(device)
    return pipe.eval()
