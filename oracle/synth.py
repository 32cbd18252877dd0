"""ORACLE -- TEST INFRASTRUCTURE ONLY.

Deterministic synthetic inputs shared by the reference run (build container), the oracle
and the HIP path: no checkpoints or datasets exist offline, so weights and
noise are *functions of a name / counter*, reproducible anywhere without files.

* `philox4x32_10`      -- Random123 Philox-4x32-10 (Salmon et al., SC'11), numpy restatement;
                          pinned by the published known-answer vectors in tests/test_oracle_cpu.py.
* `philox_normal`      -- the spec of the device noise kernel (`mf_philox_normal_f32`): element
                          quad q of sample s in draw d under `seed` = Box-Muller of
                          philox(ctr=(q, s, d, 0), key=(seed_lo, seed_hi)).  Shard-invariant by
                          construction.
* `hash_uniform`       -- counter hash (murmur3 fmix32 twice) for weights/inputs: ~20x cheaper than Philox
                          for the 194 M-parameter published UNet.
* `synth_state_dict`   -- fills every tensor of a state_dict from its *key name* so that the
                          zero-initialised tensors of the reference become non-zero
                          and the models are numerically non-degenerate.
"""
from __future__ import annotations

import zlib

import numpy as np
import torch

_M0 = np.uint64(0xD2511F53)
_M1 = np.uint64(0xCD9E8D57)
_W0 = np.uint32(0x9E3779B9)
_W1 = np.uint32(0xBB67AE85)
_MASK = np.uint64(0xFFFFFFFF)


def philox4x32_10(c0, c1, c2, c3, k0, k1):
    """Vectorised Philox-4x32-10.  All args uint32 arrays (broadcastable).  Returns 4 uint32 arrays."""
    c0, c1, c2, c3 = (np.asarray(v, dtype=np.uint32) for v in (c0, c1, c2, c3))
    k0 = np.asarray(k0, dtype=np.uint32)
    k1 = np.asarray(k1, dtype=np.uint32)
    c0, c1, c2, c3, k0, k1 = np.broadcast_arrays(c0, c1, c2, c3, k0, k1)
    with np.errstate(over="ignore"):
        for _ in range(10):
            p0 = _M0 * c0.astype(np.uint64)
            p1 = _M1 * c2.astype(np.uint64)
            hi0, lo0 = (p0 >> np.uint64(32)).astype(np.uint32), (p0 & _MASK).astype(np.uint32)
            hi1, lo1 = (p1 >> np.uint64(32)).astype(np.uint32), (p1 & _MASK).astype(np.uint32)
            c0, c1, c2, c3 = hi1 ^ c1 ^ k0, lo1, hi0 ^ c3 ^ k1, lo0
            k0 = (k0 + _W0).astype(np.uint32)
            k1 = (k1 + _W1).astype(np.uint32)
    return c0, c1, c2, c3


def _u01(x):
    """uint32 -> float32 in (0,1): ((x >> 8) + 0.5) * 2^-24 (exact in fp32)."""
    return ((x >> np.uint32(8)).astype(np.float32) + np.float32(0.5)) * np.float32(2.0 ** -24)


def philox_normal(seed: int, draw: int, sample_index: np.ndarray, per_sample: int) -> np.ndarray:
    """Standard normals, float32, shape [len(sample_index), per_sample] (per_sample % 4 == 0).

    Quad q -> elements 4q..4q+3 = (r0 cos t0, r0 sin t0, r1 cos t1, r1 sin t1) with
    r = sqrt(-2 ln u_a), t = 2 pi u_b, (u_a,u_b) = (x0,x1) and (x2,x3) of the Philox output.
    """
    assert per_sample % 4 == 0
    s = np.asarray(sample_index, dtype=np.uint32)[:, None]
    q = np.arange(per_sample // 4, dtype=np.uint32)[None, :]
    x0, x1, x2, x3 = philox4x32_10(q, s, np.uint32(draw), np.uint32(0), np.uint32(seed & 0xFFFFFFFF), np.uint32((seed >> 32) & 0xFFFFFFFF))
    two_pi = np.float32(6.283185307179586)
    out = np.empty((s.shape[0], per_sample // 4, 4), dtype=np.float32)
    for j, (a, b) in enumerate(((x0, x1), (x2, x3))):
        r = np.sqrt(np.float32(-2.0) * np.log(_u01(a)))
        t = two_pi * _u01(b)
        out[:, :, 2 * j] = r * np.cos(t)
        out[:, :, 2 * j + 1] = r * np.sin(t)
    return out.reshape(s.shape[0], per_sample)


def _fmix32(h):
    """murmur3 finaliser on uint32 arrays (in place where possible)."""
    h ^= h >> np.uint32(16)
    h *= np.uint32(0x85EBCA6B)
    h ^= h >> np.uint32(13)
    h *= np.uint32(0xC2B2AE35)
    h ^= h >> np.uint32(16)
    return h


_CHUNK = 1 << 20
_buf_u32 = np.empty(_CHUNK, dtype=np.uint32)
_buf_tmp = np.empty(_CHUNK, dtype=np.uint32)
_buf_f32 = np.empty(_CHUNK, dtype=np.float32)
_iota = np.arange(_CHUNK, dtype=np.uint32)


def hash_uniform_into(out: np.ndarray, name: str, scale: float = 1.0, shift: float = 0.0) -> None:
    """out[i] = shift + scale * u_i, u_i in (-1,1) = 24-bit mantissa of
    fmix32(fmix32(i*0x9E3779B1 + crc32(name))).  A cheap counter hash (weights must be reproducible
    and well spread, not cryptographic).  Chunked over reusable 4 MiB buffers: first-touch of fresh
    pages is the dominant cost on sandboxed hosts."""
    assert out.dtype == np.float32 and out.ndim == 1
    key = np.uint32(zlib.crc32(name.encode()))
    n = out.shape[0]
    with np.errstate(over="ignore"):
        for lo in range(0, n, _CHUNK):
            m = min(_CHUNK, n - lo)
            h, t, f = _buf_u32[:m], _buf_tmp[:m], _buf_f32[:m]
            np.add(_iota[:m], np.uint32(lo), out=h)
            h *= np.uint32(0x9E3779B1)
            h += key
            for _ in range(2):  # fmix32 twice
                np.right_shift(h, np.uint32(16), out=t); h ^= t
                h *= np.uint32(0x85EBCA6B)
                np.right_shift(h, np.uint32(13), out=t); h ^= t
                h *= np.uint32(0xC2B2AE35)
                np.right_shift(h, np.uint32(16), out=t); h ^= t
            np.right_shift(h, np.uint32(8), out=t)
            f[:] = t                      # exact: < 2^24
            f += np.float32(0.5)
            f *= np.float32(2.0 ** -23)   # ((x>>8)+0.5) * 2^-24 * 2
            f -= np.float32(1.0)
            if scale != 1.0:
                f *= np.float32(scale)
            if shift != 0.0:
                f += np.float32(shift)
            out[lo:lo + m] = f


def hash_uniform(name: str, n: int) -> np.ndarray:
    out = np.empty(n, dtype=np.float32)
    hash_uniform_into(out, name)
    return out


philox_uniform = hash_uniform  # name kept for callers; weights/inputs use the hash, noise uses Philox


def _scale_shift(shape, kind: str):
    """kind: 'weight' (uniform * sqrt(3/fan_in)), 'bias' (0.1*uniform), 'gamma' (1 + 0.2*uniform),
    'embedding' (uniform)."""
    if kind == "weight":
        fan_in = int(np.prod(shape[1:])) if len(shape) > 1 else int(shape[0])
        return float(np.float32(np.sqrt(3.0 / max(fan_in, 1)))), 0.0
    if kind == "bias":
        return 0.1, 0.0
    if kind == "gamma":
        return 0.2, 1.0
    if kind == "embedding":
        return 1.0, 0.0
    raise ValueError(kind)


def synth_tensor(name: str, shape, kind: str) -> torch.Tensor:
    out = torch.empty(tuple(shape), dtype=torch.float32)
    sc, sh = _scale_shift(tuple(shape), kind)
    hash_uniform_into(out.view(-1).numpy(), name, sc, sh)
    return out


def _kind_of(key: str, shape) -> str:
    leaf = key.rsplit(".", 1)[-1]
    if "embedding" in key and leaf == "weight" and len(shape) == 2 and ".local_embedder" not in key and "time_emb" not in key:
        return "embedding"
    if leaf == "weight":
        return "gamma" if len(shape) == 1 else "weight"
    if leaf == "bias":
        return "bias"
    if leaf == "weights" and len(shape) == 1:   # LearnedSinusoidalPosEmb's frequency vector (time_embedder.py:39): randn-like spread
        return "embedding"
    raise ValueError(f"unexpected parameter {key}")


@torch.no_grad()
def synth_state_dict(module: torch.nn.Module, prefix: str = "") -> dict:
    """Overwrite every *parameter* of `module` in place from its key name; buffers untouched.
    Returns the new state_dict.  The same call on the reference module, the oracle module and
    the product module yields identical weights because the keys are identical."""
    for key, p in module.named_parameters():
        shape = tuple(p.shape)
        sc, sh = _scale_shift(shape, _kind_of(key, shape))
        if p.device.type == "cpu" and p.is_contiguous():
            hash_uniform_into(p.data.view(-1).numpy(), prefix + key, sc, sh)
        else:
            p.copy_(synth_tensor(prefix + key, shape, _kind_of(key, shape)))
    return module.state_dict()


def synth_input(name: str, shape, scale: float = 1.0) -> torch.Tensor:
    out = torch.empty(tuple(shape), dtype=torch.float32)
    hash_uniform_into(out.view(-1).numpy(), "input:" + name, scale * 1.7320508)
    return out


class PhiloxNoise:
    """Host-side noise source in reference draw order: call #d returns draw d."""

    def __init__(self, seed: int, sample_offset: int = 0):
        self.seed, self.draw, self.sample_offset = seed, 0, sample_offset

    def __call__(self, like: torch.Tensor) -> torch.Tensor:
        b = like.shape[0]
        per = int(np.prod(like.shape[1:]))
        idx = np.arange(self.sample_offset, self.sample_offset + b)
        out = philox_normal(self.seed, self.draw, idx, per).reshape(tuple(like.shape))
        self.draw += 1
        return torch.from_numpy(out).to(like.device)
