"""Shared test helpers.  The oracle (oracle/) is the CHECKER here, never the thing under test on GPU runs."""
from __future__ import annotations

from pathlib import Path

import numpy as np
import torch

from oracle import restate as R
from oracle import synth as S

GOLD = Path(__file__).resolve().parent / "golden"


def gold(name: str) -> dict:
    with np.load(GOLD / f"{name}.npz") as z:
        return {k: z[k] for k in z.files}


def T(a) -> torch.Tensor:
    return torch.from_numpy(np.asarray(a))


def relerr(a: torch.Tensor, b: torch.Tensor) -> float:
    """max-norm relative error: max|a-b| / max|b| (the metric of SURVEY §8d)."""
    a, b = a.detach().cpu().double(), b.detach().cpu().double()
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))


def relerr_rows(a: torch.Tensor, b: torch.Tensor) -> float:
    """max over the samples (leading axis) of the PER-SAMPLE max-norm relative error max|a_n - b_n| / max|b_n|: the whole-tensor figure of
    `relerr` divides by the largest sample of the batch, so an error confined to a small-magnitude sample hides behind a large one
    (VERDICT r03, weak 1)."""
    a, b = a.detach().cpu().double(), b.detach().cpu().double()
    a, b = a.reshape(a.shape[0], -1), b.reshape(b.shape[0], -1)
    return float(((a - b).abs().amax(1) / b.abs().amax(1).clamp_min(1e-30)).max())


def relerr_rms(a: torch.Tensor, b: torch.Tensor) -> float:
    """max over the samples of the per-sample RMS-relative error sqrt(mean (a_n - b_n)^2 / mean b_n^2): an error spread over many small
    elements shows here although no single element moves the max norm."""
    a, b = a.detach().cpu().double(), b.detach().cpu().double()
    a, b = a.reshape(a.shape[0], -1), b.reshape(b.shape[0], -1)
    return float(((a - b).pow(2).mean(1) / b.pow(2).mean(1).clamp_min(1e-60)).sqrt().max())


def to_product_kwargs(kw: dict) -> dict:
    """oracle kwargs -> product kwargs (swap the embedder classes)."""
    import medfusion_amd as M

    kw = dict(kw)
    if kw.get("time_embedder") is R.TimeEmbbeding:
        kw["time_embedder"] = M.TimeEmbbeding
    if kw.get("cond_embedder") is R.LabelEmbedder:
        kw["cond_embedder"] = M.LabelEmbedder
    return kw


def oracle_fp64_drift(ora, seed: int, *args, **kw) -> float:
    """How far the fp32 ORACLE lands from its own fp64 self on one sample() case: the conditioning of the case, measured.  A case whose fp32
    evaluation is only determined to `drift` cannot be held to less by any fp32-class implementation; tests whose case is ill-conditioned
    derive their bound from this figure instead of a hand-widened constant.  The noise fn of `ora` is left re-seeded."""
    import copy

    class _Noise64:
        def __init__(self, s):
            self.f = S.PhiloxNoise(s)

        def __call__(self, like):
            return self.f(like).double()

    o64 = copy.deepcopy(ora).double()
    ora.set_noise_fn(S.PhiloxNoise(seed))
    w32 = ora.sample(*args, **kw)
    o64.set_noise_fn(_Noise64(seed))
    torch.set_default_dtype(torch.float64)   # (the restatement builds its sinusoidal tables / scalars in the default dtype)
    try:
        w64 = o64.sample(*args, **kw)
    finally:
        torch.set_default_dtype(torch.float32)
    ora.set_noise_fn(S.PhiloxNoise(seed))
    assert w64.dtype == torch.float64
    return relerr(w32, w64)


def oracle_noise(seed: int):
    """HostNoise that replays the oracle's numpy-Philox draws (same draws the golden fixtures were made with)."""
    import medfusion_amd as M

    src = S.PhiloxNoise(seed)
    return M.HostNoise(lambda shape: src(torch.empty(tuple(shape))))


# (N, H, W, C1, C2, Cout of both, tile / split-K hint of the 3x3 (0: planner), tile hint of the 1x1 (0: planner))
GROUP_CASES = [(16, 16, 16, 256, 0, 512, 0, 0, 0), (16, 8, 8, 512, 0, 1024, 0, 0, 0), (16, 8, 8, 1024, 1024, 1024, 0, 0, 36), (16, 8, 8, 1024, 512, 512, 0, 0, 36),
               (16, 16, 16, 512, 512, 512, 0, 0, 36), (16, 16, 16, 512, 256, 256, 0, 0, 36), (16, 32, 32, 256, 256, 256, 0, 0, 37),
               (2, 16, 16, 64, 0, 128, 53, 1, 53), (3, 16, 16, 64, 32, 128, 54, 2, 53), (2, 16, 16, 128, 0, 256, 34, 2, 37), (2, 32, 32, 64, 64, 128, 62, 1, 36),
               (3, 16, 16, 96, 0, 128, 34, 1, 36)]
