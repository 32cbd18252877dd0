"""Multi-GPU sharding of `sample()`: samples are independent, so the batch rows are partitioned
across one process per GPU with NO collective inside the denoise loop; exactly one all-gather of the decoded
images at the end (torch.distributed backend "nccl" == RCCL over xGMI on ROCm; "gloo" in CPU tests).
The reference has no distributed code at all -- this is new functionality, not a port.
"""
from __future__ import annotations

import os
from typing import Optional, Tuple

import torch
import torch.distributed as dist


def shard_rows(n: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous, balanced partition of n rows: the first n % world ranks get one extra row."""
    if not (0 <= rank < world):
        raise ValueError(f"rank {rank} not in [0, {world})")
    q, r = divmod(n, world)
    lo = rank * q + min(rank, r)
    return lo, lo + q + (1 if rank < r else 0)


def init_from_env(backend: Optional[str] = None) -> Tuple[int, int, int]:
    """Reads RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* (torch.distributed.run contract).  Returns (rank, local_rank, world)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    force = os.environ.get("MEDFUSION_FORCE_COLLECTIVE", "0") == "1"   # a 1-rank group too: the collective path on the hardware at hand
    if (world > 1 or force) and not dist.is_initialized():
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        if backend == "nccl":
            torch.cuda.set_device(local)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, local, world


def gather_images(local: torch.Tensor, num_samples: int, rank: int, world: int) -> torch.Tensor:
    """All-gather the per-rank image shards [B_r, C, H, W] into the global [num_samples, C, H, W] (row order = sample index).
    Shards may differ by one row; they are padded to the largest and trimmed after the collective.
    world == 1 returns the shard as it is -- unless MEDFUSION_FORCE_COLLECTIVE=1 and a process group exists: then the 1-rank group runs
    the same all-gather (on device memory over RCCL when the backend is nccl), which is how the collective path is exercised on a
    single-GPU box (tests/test_multiproc_gpu.py::test_rccl_world1_device_allgather)."""
    if world == 1 and not (os.environ.get("MEDFUSION_FORCE_COLLECTIVE", "0") == "1" and dist.is_initialized()):
        return local
    sizes = [shard_rows(num_samples, r, world) for r in range(world)]
    mx = max(hi - lo for lo, hi in sizes)
    pad = local
    if local.shape[0] < mx:
        pad = torch.cat([local, local.new_zeros((mx - local.shape[0], *local.shape[1:]))], dim=0)
    pad = pad.contiguous()
    host = dist.get_backend() == "gloo" and pad.is_cuda   # gloo (CPU tests, single-GPU boxes) gathers host tensors; nccl == RCCL moves device memory
    src = pad.cpu() if host else pad
    bufs = [torch.empty_like(src) for _ in range(world)]
    dist.all_gather(bufs, src)
    out = torch.cat([b[: hi - lo] for b, (lo, hi) in zip(bufs, sizes)], dim=0)
    return out.to(local.device) if host else out


def sample_sharded(pipeline, num_samples, img_size, condition=None, noise=None, gather=True, **kwargs) -> torch.Tensor:
    """`pipeline.sample` over all ranks of the default process group; returns the full batch on every rank."""
    rank = dist.get_rank() if dist.is_initialized() else 0
    world = dist.get_world_size() if dist.is_initialized() else 1
    local = pipeline.sample(num_samples, img_size, condition=condition, noise=noise, shard=(rank, world), **kwargs)
    return gather_images(local, num_samples, rank, world) if gather else local
