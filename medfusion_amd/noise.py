"""Noise sources for the sampling path.

The reference draws `torch.randn_like` on the default generator of the tensor's device, in a fixed order
(Q3): one draw in `sample` (x_T), then per loop iteration the posterior draw (gaussian_scheduler.py:99,
always) and, on all but the last DDIM iteration, the DDIM draw (diffusion_pipeline.py:303).  A NoiseSource
reproduces exactly that *sequence of draws* and is shard-invariant: rank r asks for rows
[offset, offset+B) of the GLOBAL batch.

* PhiloxDeviceNoise -- counter-based generator on the GPU (mf_philox_normal_f32); graph-capturable; default.
                       Without an explicit seed every `begin()` takes a fresh 64-bit key from torch's default CPU generator
                       (and so ADVANCES it, like the reference's randn_like advances its generator): successive `sample()` /
                       `forward()` / `encode()` calls differ, `torch.manual_seed(s)` before a call reproduces it, and ranks that
                       seed identically derive identical keys (shard invariance).  With `seed=` the key is fixed: the same
                       call repeated gives the same draws (parity tests, bench steps use distinct seeds).
* HostNoise         -- wraps any host callable `fn(shape) -> CPU tensor` producing the GLOBAL-batch draw
                       (e.g. torch's CPU generator, or the oracle's numpy Philox) and uploads the rank's rows:
                       used by parity tests to inject bit-identical noise into oracle and HIP path.
"""
from __future__ import annotations

from typing import Callable, Optional, Sequence

import torch

from . import kernels as K


class NoiseSource:
    def begin(self, local_batch: int, device, sample_offset: int = 0, global_batch: Optional[int] = None) -> None:
        self.local_batch, self.device, self.sample_offset = local_batch, device, sample_offset
        self.global_batch = local_batch if global_batch is None else global_batch
        self.draw_index = 0

    def draw(self, shape: Sequence[int], out: Optional[torch.Tensor] = None) -> torch.Tensor:
        raise NotImplementedError


class PhiloxDeviceNoise(NoiseSource):
    def __init__(self, seed: Optional[int] = None):
        self.seed = seed

    def begin(self, local_batch, device, sample_offset=0, global_batch=None):
        super().begin(local_batch, device, sample_offset, global_batch)
        if self.seed is None:
            # stateful like the reference's default generator: the key comes out of torch's CPU generator, which moves on
            self._seed = int(torch.randint(0, 2 ** 62, (1,), dtype=torch.int64).item())
        else:
            self._seed = int(self.seed)

    def draw(self, shape, out=None):
        if out is None:
            out = torch.empty(tuple(shape), dtype=torch.float32, device=self.device)
        K.philox_normal(out, self._seed, self.draw_index, self.sample_offset)
        self.draw_index += 1
        return out

    def draw_indexed(self, out: torch.Tensor, draw_base: int, draw_stride: int, step_dev: torch.Tensor) -> torch.Tensor:
        """draw number = draw_base + draw_stride * (*step_dev): replayable inside a captured graph."""
        return K.philox_normal(out, self._seed, draw_base, self.sample_offset, step_dev=step_dev, draw_stride=draw_stride)


class HostNoise(NoiseSource):
    def __init__(self, fn: Callable[[Sequence[int]], torch.Tensor]):
        self.fn = fn

    def draw(self, shape, out=None):
        gshape = (self.global_batch, *shape[1:])
        full = self.fn(gshape)
        assert tuple(full.shape) == tuple(gshape), (full.shape, gshape)
        rows = full[self.sample_offset:self.sample_offset + self.local_batch].to(torch.float32).contiguous()
        self.draw_index += 1
        if out is None:
            return rows.to(self.device)
        out.copy_(rows)
        return out


def torch_cpu_noise(seed: Optional[int] = None) -> HostNoise:
    """Draws from a torch CPU generator exactly like the reference running on CPU (`torch.manual_seed(seed)`)."""
    gen = None
    if seed is not None:
        gen = torch.Generator(device="cpu")
        gen.manual_seed(seed)
    return HostNoise(lambda shape: torch.randn(tuple(shape), generator=gen))


def default_noise() -> NoiseSource:
    return PhiloxDeviceNoise()
