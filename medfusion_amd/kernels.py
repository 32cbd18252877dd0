"""Tensor-level wrappers over the C-ABI.  torch is plumbing only: device memory, streams.

Activations inside the path are NHWC torch tensors of shape [N, H, W, C] (contiguous).  Every wrapper
launches on torch's current stream and raises if a tensor is not on the GPU -- there is no CPU path.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

import torch

from . import lib as L


def _gpu(*ts):
    for t in ts:
        if t is None:
            continue
        if not t.is_cuda:
            raise RuntimeError("medfusion_amd: tensors must live on a ROCm device -- the product path has no CPU fallback")
        if t.dtype not in (torch.float32, torch.float64, torch.int64, torch.int32, torch.uint8, torch.int16):
            raise RuntimeError(f"medfusion_amd: unsupported dtype {t.dtype}")


def _ptr(t: Optional[torch.Tensor]) -> Optional[int]:
    return None if t is None else t.data_ptr()


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)
_cur_device = getattr(torch._C, "_cuda_getDevice", None)


def stream(device_index: Optional[int] = None) -> int:
    """the hipStream_t torch launches on right now (current device, or `device_index`): the raw C entry points when this torch has them --
    `torch.cuda.current_stream()` builds a Stream object per call, a quarter of the host time of a launch (scripts/host_profile.py)"""
    if _raw_stream is not None and _cur_device is not None:
        return _raw_stream(_cur_device() if device_index is None else device_index)
    return (torch.cuda.current_stream() if device_index is None else torch.cuda.current_stream(device_index)).cuda_stream


_growths = 0   # how often a Workspace / SyncWords buffer was (re)allocated: the command-list recorder compares before / after (pipeline.py)


def scratch_growth_count() -> int:
    return _growths


class Workspace:
    """Grow-only scratch owned by torch (the library never allocates).  One per (device, stream): launches on one
    stream are serialised so sharing is safe.  Grown only outside graph capture (warm-up run does it)."""

    _bufs = {}

    @classmethod
    def get(cls, nbytes: int, device) -> torch.Tensor:
        key = (device.index, stream(device.index))  # per stream: concurrent streams must not share scratch
        buf = cls._bufs.get(key)
        if buf is None or buf.numel() < nbytes:
            if torch.cuda.is_current_stream_capturing():
                raise RuntimeError("medfusion_amd: workspace growth during graph capture; run one eager warm-up first")
            buf = torch.empty(max(nbytes, 1 << 20), dtype=torch.uint8, device=device)
            cls._bufs[key] = buf
            global _growths
            _growths += 1
        return buf


class SyncWords:
    """Zero-initialised 32-bit counters the in-launch split-K reduction of mf_conv2d_f16x2 meets through (every launch leaves them zero).
    One array per (device, stream), like the workspace; grown (re-zeroed) outside graph capture only."""

    _bufs = {}

    @classmethod
    def get(cls, words: int, device) -> torch.Tensor:
        key = (device.index, stream(device.index))
        buf = cls._bufs.get(key)
        if buf is None or buf.numel() < words:
            if torch.cuda.is_current_stream_capturing():
                raise RuntimeError("medfusion_amd: sync-counter growth during graph capture; run one eager warm-up first")
            buf = torch.zeros(max(words, 1 << 14), dtype=torch.int32, device=device)
            cls._bufs[key] = buf
            global _growths
            _growths += 1
        return buf

    @classmethod
    def reset(cls, device) -> None:
        """Zero the counters of the current stream again (one memset): a launch that died half-way -- a device fault, an interrupted
        process -- would otherwise leave a pair's counter at 1 and the next launch would take a stale tile.  Called once per sampling loop /
        VAE pass, never per launch."""
        buf = cls._bufs.get((device.index, stream(device.index)))
        if buf is not None and not torch.cuda.is_current_stream_capturing():
            buf.zero_()


class Rendezvous:
    """Counters of the convolutions that apply their GroupNorm inside their own launch (mf_conv2d_f16x2_gn_apply): word 0 is the ERROR FLAG
    (a wait that did not complete: another process's waiting workgroups filled the device), the [N][2] arrive / depart counters start at
    word 4.  One array per (device, stream); zero before the first launch, left zero by every launch.  `failed(device)` reads the flags of
    the device (a host sync): DiffusionPipeline checks once per sampling loop, VAE once per pass, and on a hit zero everything, switch the
    fused form off for the process (`disabled`) and re-run on the two-launch form."""

    _bufs = {}
    # OPT-IN (MEDFUSION_FUSED_APPLY=1): built, bit-identical to the two-launch form, and measured NOT faster -- same-process A/B on cfg2,
    # profiles/r04_fused_gn_apply_ab.txt: -0.2 % (fused at 32 x 32 only) ... -2.2 % (everywhere); the tail costs what the boundary + the
    # apply launch cost (profiles/r04_conv_timeline_fused.txt).  So the default is the two-launch form, which also never waits in a kernel.
    disabled = os.environ.get("MEDFUSION_FUSED_APPLY", "0") != "1"
    used = {}          # device index -> a fused launch went out since the last check
    launches = 0       # fused launches issued by this process (tests, diagnostics)

    @classmethod
    def get(cls, words: int, device) -> torch.Tensor:
        key = (device.index, stream(device.index))
        buf = cls._bufs.get(key)
        if buf is None or buf.numel() < words + 4:
            if torch.cuda.is_current_stream_capturing():
                raise RuntimeError("medfusion_amd: rendezvous-counter growth during graph capture; run one eager warm-up first")
            buf = torch.zeros(max(words + 4, 1 << 12), dtype=torch.int32, device=device)
            cls._bufs[key] = buf
            global _growths
            _growths += 1
        cls.used[device.index] = True
        cls.launches += 1
        return buf

    @classmethod
    def failed(cls, device) -> bool:
        """did a rendezvous of this device time out since the last check?  (host sync; clears flag and counters on a hit)"""
        if not cls.used.get(device.index):
            return False
        cls.used[device.index] = False
        torch.cuda.synchronize(device)   # (every stream of the device: .item() below would wait for the current one only)
        bad = False
        for (di, _), buf in list(cls._bufs.items()):
            if di == device.index and int(buf[0].item()) != 0:
                bad = True
        if bad:
            for (di, _), buf in list(cls._bufs.items()):
                if di == device.index:
                    buf.zero_()
        return bad


# Policy (blocks.py): a convolution applies its GroupNorm itself only from this many output pixels per sample on.  Below, the tiles of a
# sample finish too far apart (deep in-launch split-K trees at 8 x 8) for the wait to beat a separate apply launch (scripts/conv_timeline.py
# --fused, profiles/r04_conv_timeline_fused.txt).  The C-ABI query (mf_conv2d_f16x2_fuse_words) answers capability only.
FUSE_MIN_HW = int(os.environ.get("MEDFUSION_FUSE_MIN_HW", "256"))
_fused = __import__("threading").local()   # .depth: nesting of with_fused_fallback on THIS thread (two threads may drive two streams)


def with_fused_fallback(device, fn, rewind=None):
    """Run fn() -- a whole sampling loop, a VAE pass -- and, if one of the convolutions that apply their GroupNorm inside their own launch
    timed out waiting for its sample (Rendezvous: only when another process's waiting workgroups fill the device), switch that form off for
    the process, call `rewind()` (restore whatever fn consumed: noise counters) and run fn() again on the two-launch form.  Nested calls
    (the decode inside a sampling loop) leave the check to the outermost one: a failure anywhere invalidates everything after it."""
    if getattr(_fused, "depth", 0) > 0 or Rendezvous.disabled or torch.cuda.is_current_stream_capturing():
        return fn()
    _fused.depth = 1
    try:
        out = fn()
    finally:
        _fused.depth = 0
    if Rendezvous.failed(device):
        import warnings
        Rendezvous.disabled = True
        warnings.warn("medfusion_amd: a convolution's in-launch GroupNorm rendezvous timed out (the device is shared with another process whose "
                      "workgroups were waiting too); the fused form is switched off for this process and the pass is re-run on the two-launch form")
        if rewind is not None:
            rewind()
        SyncWords.reset(device)
        out = fn()
    return out


def pack_conv_weight(w_oihw: torch.Tensor) -> torch.Tensor:
    """OIHW -> [Cout][KH][KW][Cin] (device, once at load)."""
    _gpu(w_oihw)
    w = w_oihw.contiguous()
    co, ci, kh, kw = w.shape
    out = torch.empty((co, kh, kw, ci), dtype=torch.float32, device=w.device)
    L.check(L.load().mf_pack_conv_weight_f32(w.data_ptr(), out.data_ptr(), co, ci, kh, kw, stream()), "mf_pack_conv_weight_f32")
    return out


def pack_upconv_weight(w_oihw: torch.Tensor) -> torch.Tensor:
    """OIHW 3x3 -> [4 phases][Cout][2][2][Cin]: weights of the sub-pixel form of nearest-x2 + 3x3 conv."""
    _gpu(w_oihw)
    w = w_oihw.contiguous()
    co, ci, kh, kw = w.shape
    assert (kh, kw) == (3, 3)
    out = torch.empty((4, co, 2, 2, ci), dtype=torch.float32, device=w.device)
    L.check(L.load().mf_pack_upconv_weight_f32(w.data_ptr(), out.data_ptr(), co, ci, stream()), "mf_pack_upconv_weight_f32")
    return out


def subpixel_ok(d: L.MfConvDesc) -> bool:
    return bool(L.load().mf_conv2d_subpixel_ok(C.byref(d)))


def conv_is_igemm(d: L.MfConvDesc) -> bool:
    return bool(L.load().mf_conv2d_is_igemm(C.byref(d)))


def split_conv_weight(w_packed: torch.Tensor) -> torch.Tensor:
    """packed fp32 weights (either packing; rows = all leading dims but the K = kh*kw*cin of the last three) -> the bf16-triplet
    form of MF_CONV_FP32_SPLIT3_W3: [rows][K/8][3][8] bf16 as an opaque int16 tensor [rows, 3K]."""
    _gpu(w_packed)
    w = w_packed.contiguous()
    k = w.shape[-1] * w.shape[-2] * w.shape[-3]
    rows = w.numel() // k
    out = torch.empty((rows, 3 * k), dtype=torch.int16, device=w.device)
    L.check(L.load().mf_split_conv_weight_bf16x3(w.data_ptr(), out.data_ptr(), rows, k, stream()), "mf_split_conv_weight_bf16x3")
    return out


# ----------------------------------------------------------------------------- fp16-pair operands (MF_CONV_FP32_F16X2)
# The fp16-pair form of an activation travels as attributes of the fp32 tensor it mirrors: `t._mf_split` (an int32 tensor of the same
# shape) and `t._mf_bound` (float [N]: the per-sample upper bound of |t| whose exponent scaled the pairs -- fp16 stops at 65504, an
# un-normalised residual stream does not; csrc/split_f16.h).  Producers that know a bound for free set both (gn_apply(split=True)); the
# fp16-pair convolution can MEASURE the bound of its output (measure_out=True); a consumer that finds neither runs the stand-alone
# passes (max |t| per sample, then the split) once and caches the result on the tensor.  Every wrapper that writes INTO an existing
# tensor (`out=`) drops stale mirrors first.
# Bound-slack audit (diagnostic, MEDFUSION_AUDIT_BOUNDS=1 or kernels.AUDIT = True; tests/test_parity_gpu.py::test_trained_like_weights_and_bound_slack):
# every producer that attaches an fp16-pair mirror under a DERIVED bound records (site, shape, bound / true max per sample) in AUDIT_LOG.  A derived
# bound sits above the data (the pair format keeps its 23 bits down to 2^-28 of the bound); the audit shows by how much, across a whole network.
AUDIT = os.environ.get("MEDFUSION_AUDIT_BOUNDS", "0") == "1"
AUDIT_LOG = []


def _audit(t: torch.Tensor, site: str) -> None:
    """diagnostic only (torch arithmetic, a host sync): decode the pair mirror of `t`, compare its per-sample max with the bound it was scaled by"""
    if not AUDIT or getattr(t, "_mf_split", None) is None or getattr(t, "_mf_bound", None) is None or torch.cuda.is_current_stream_capturing():
        return
    n = t.shape[0]
    raw = t._mf_split.view(torch.float16).reshape(n, -1, 2, 8).float()
    bound = t._mf_bound.double()
    s = torch.floor(torch.log2(bound.clamp_min(1e-300))) - 14
    amax = (raw[:, :, 0] + raw[:, :, 1] / 2048.0).abs().amax(dim=(1, 2)).double() * torch.exp2(s)
    slack = (bound / amax.clamp_min(1e-300)).cpu()
    AUDIT_LOG.append((site, tuple(t.shape), float(slack.max()), float(slack.min()), bool(torch.isfinite(raw).all())))


def _stamp(t: torch.Tensor) -> None:
    """remember the tensor's version counter next to its mirrors: a torch in-place op on `t` bumps `_version`, and stale() sees it (writes
    through this library go by raw pointer and do not: every wrapper that writes INTO an existing tensor calls drop_split itself)"""
    t._mf_ver = t._version


def stale(t: torch.Tensor) -> bool:
    return getattr(t, "_mf_ver", None) is not None and t._mf_ver != t._version


def _fresh(t: torch.Tensor, name: str):
    """attribute `name` of t (a mirror), or None -- dropping every mirror first when a torch in-place op has touched t since they were made"""
    if stale(t):
        if getattr(t, "_mf_pairs_only", False):
            # the pairs are the ONLY valid copy of this tensor (its fp32 storage was never written): a torch in-place op on it has just
            # modified garbage, and dropping the mirror would make every later reader take that garbage for the values
            raise RuntimeError("medfusion_amd: a torch in-place op touched a tensor that exists only as fp16 pairs (the output of a GroupNorm-apply "
                               "pass between two convolutions); its fp32 storage holds no values")
        drop_split(t)
        t._mf_ver = None
    return getattr(t, name, None)


def pairs_only(t) -> bool:
    """this tensor's values exist only in its fp16-pair mirror (gn_apply(out_fp32=False)): its fp32 storage was never written"""
    return t is not None and getattr(t, "_mf_pairs_only", False)


def _need_f32(*ts):
    for t in ts:
        if pairs_only(t):
            raise RuntimeError("medfusion_amd: this tensor exists only as fp16 pairs (the output of a GroupNorm-apply pass between two convolutions); "
                               "its fp32 form was never written -- only the fp16-pair convolution and the apply pass (as a residual) can read it")


def drop_split(t: Optional[torch.Tensor]) -> None:
    if t is not None:
        if getattr(t, "_mf_pairs_only", False):
            t._mf_pairs_only = False   # (an fp32 writer is about to fill the storage)
        if getattr(t, "_mf_split", None) is not None:
            t._mf_split = None
        if getattr(t, "_mf_bound", None) is not None:
            t._mf_bound = None
        if getattr(t, "_mf_slots", None) is not None:
            t._mf_slots = None
        if getattr(t, "_mf_wino", None) is not None:
            t._mf_wino = t._mf_wino_bound = None
        if getattr(t, "_mf_wino_f32", None) is not None:
            t._mf_wino_f32 = None


def maxabs_rows(x: torch.Tensor) -> torch.Tensor:
    """x [N, ...] contiguous fp32 -> bound [N] = max |x[n]| (measured on the device: two small launches, no atomics, no host sync)"""
    _gpu(x)
    _need_f32(x)
    if x.dtype != torch.float32 or not x.is_contiguous():
        raise RuntimeError("maxabs_rows: contiguous fp32")
    n = x.shape[0]
    lib = L.load()
    per = x.numel() // n
    partial = torch.empty((n, lib.mf_maxabs_rows_slots(per)), dtype=torch.float32, device=x.device)
    bound = torch.empty((n,), dtype=torch.float32, device=x.device)
    L.check(lib.mf_maxabs_rows_f32(x.data_ptr(), partial.data_ptr(), bound.data_ptr(), n, per, stream()), "mf_maxabs_rows_f32")
    return bound


def bound_of(x: torch.Tensor) -> torch.Tensor:
    b = _fresh(x, "_mf_bound")
    if b is None:
        sl = _fresh(x, "_mf_slots")
        if sl is not None:   # per-(tile, wave) maxima the producing convolution left: one wave per sample reduces them
            b = torch.empty((x.shape[0],), dtype=torch.float32, device=x.device)
            L.check(L.load().mf_bound_finalize_f32(sl.data_ptr(), b.data_ptr(), sl.shape[0], sl.shape[1], stream()), "mf_bound_finalize_f32")
        else:
            b = maxabs_rows(x)
        x._mf_bound = b
        _stamp(x)
    return b


def split_f16x2(x: torch.Tensor, bound: Optional[torch.Tensor] = None) -> torch.Tensor:
    """fp32 tensor [N, ...] (innermost extent % 8 == 0) -> its fp16-pair form, an opaque int32 tensor of the same shape; row n is scaled
    by the power of two that `bound[n]` implies (None: unscaled)"""
    _gpu(x, bound)
    _need_f32(x)
    if x.dtype != torch.float32 or not x.is_contiguous() or x.shape[-1] % 8:
        raise RuntimeError("split_f16x2: contiguous fp32 with innermost extent % 8 == 0")
    rows = x.shape[0] if bound is not None else 1
    if bound is not None and bound.numel() != rows:
        raise RuntimeError("split_f16x2: one bound per leading index")
    out = torch.empty(x.shape, dtype=torch.int32, device=x.device)
    L.check(L.load().mf_split_f16x2(x.data_ptr(), out.data_ptr(), _ptr(bound), rows, x.numel() // rows, stream()), "mf_split_f16x2")
    return out


def split_of(x: torch.Tensor) -> torch.Tensor:
    s = _fresh(x, "_mf_split")
    if s is None:
        if _fresh(x, "_mf_bound") is None and x.dim() >= 2 and x.shape[-1] % 8 == 0 and x.is_contiguous() and x.dtype == torch.float32:
            # no bound yet: the slot maxima the producing convolution left -- or those of a measuring pass -- are reduced inside the split
            # itself (mf_split_f16x2_slots: no bound-finalize launch), which publishes the bound as well
            _need_f32(x)
            lib = L.load()
            n, per = x.shape[0], x.numel() // x.shape[0]
            sl = _fresh(x, "_mf_slots")
            if sl is None:
                sl = torch.empty((n, lib.mf_maxabs_rows_slots(per)), dtype=torch.float32, device=x.device)
                L.check(lib.mf_maxabs_rows_f32(x.data_ptr(), sl.data_ptr(), None, n, per, stream()), "mf_maxabs_rows_f32")
            s = torch.empty(x.shape, dtype=torch.int32, device=x.device)
            b = torch.empty((n,), dtype=torch.float32, device=x.device)
            L.check(lib.mf_split_f16x2_slots(x.data_ptr(), s.data_ptr(), sl.data_ptr(), sl.shape[1], b.data_ptr(), n, per, stream()), "mf_split_f16x2_slots")
            x._mf_bound = b
        else:
            s = split_f16x2(x, bound_of(x))
        x._mf_split = s
        _stamp(x)
        _audit(x, "split of a measured tensor")
    return s


def split_weight_f16x2(w_packed: torch.Tensor):
    """packed fp32 conv weights -> (fp16-pair form scaled by their own max, that max as a float); load-time work (one host sync)"""
    if torch.cuda.is_current_stream_capturing():
        #  the split needs max |w| on the HOST: a weight tensor whose first use falls inside a graph capture -- a shape first seen there
        # -- cannot be packed; every loop form runs its first iteration eagerly for exactly this reason
        raise RuntimeError("medfusion_amd: conv weights must be packed (a host sync) before a hipGraph capture starts: run one eager evaluation of this "
                           "input shape first (DiffusionPipeline.denoise does: its iteration 0 is eager in every loop form)")
    w = w_packed.contiguous()
    wmax = float(w.abs().max().item())
    bound = torch.full((1,), wmax, dtype=torch.float32, device=w.device)
    return split_f16x2(w.view(1, -1), bound).view(w.shape), wmax


def conv_f16x2_ok(d: L.MfConvDesc) -> bool:
    return bool(L.load().mf_conv2d_f16x2_ok(C.byref(d)))


def conv_plan(d: L.MfConvDesc):
    """(tile id, split-K factor) the planner picks for `d`; (0, 0) if it is not on an implicit-GEMM kernel"""
    t, k = C.c_int32(), C.c_int32()
    L.check(L.load().mf_conv2d_plan_query(C.byref(d), C.byref(t), C.byref(k)), "mf_conv2d_plan_query")
    return t.value, k.value


def pin_conv_plan(d: L.MfConvDesc):
    """Fix the planner's choice for `d` in its hint fields (later calls with this descriptor skip the table lookup and the cost model) and
    return what a caller needs per launch: (workspace bytes, slots of the measured-bound array, sync words of an in-launch split-K)."""
    lib = L.load()
    if d.precision in (5, 6) and not (d.tile_hint and d.splitk_hint):
        t, k = conv_plan(d)
        if t > 0 and k > 0:
            d.tile_hint, d.splitk_hint = t, k
    return (lib.mf_conv2d_workspace_bytes(C.byref(d)), (lib.mf_conv2d_f16x2_bound_slots(C.byref(d)) if d.precision in (5, 6) else 0),
            (lib.mf_conv2d_f16x2_sync_words(C.byref(d)) if d.precision in (5, 6) else 0))


def conv2d_f16x2(x1: torch.Tensor, w_split, bias: Optional[torch.Tensor], d: L.MfConvDesc, x2: Optional[torch.Tensor] = None,
                 out: Optional[torch.Tensor] = None, measure_out: bool = False, gn_groups: int = 0, gn_parts: int = 0, pinned=None):
    """MF_CONV_FP32_F16X2 (d.precision = 5; or the opt-in single-term MF_CONV_F16 = 6) convolution of fp32 NHWC tensors whose fp16-pair
    mirrors are made on demand.  w_split = split_weight_f16x2(...).
    measure_out: also measure the per-sample max |y| (-> y._mf_bound: the output feeds a convolution or a residual add un-normalised).
    pinned: pin_conv_plan(d), computed once by callers that launch the same descriptor every iteration.
    Returns y, or (y, partial [N, parts, G, 2]) when gn_groups > 0 (statistics of the GroupNorm that follows)."""
    wh, wmax = w_split
    _gpu(x1, x2, wh, bias)
    lib = L.load()
    x1s, b1 = split_of(x1), bound_of(x1)
    x2s, b2 = (split_of(x2), bound_of(x2)) if x2 is not None else (None, None)
    ho, wo = conv_out_hw(d)
    if out is None:
        out = torch.empty((d.N, ho, wo, d.Cout), dtype=torch.float32, device=x1.device)
    else:
        drop_split(out)
    if pinned is None:
        pinned = (lib.mf_conv2d_workspace_bytes(C.byref(d)), lib.mf_conv2d_f16x2_bound_slots(C.byref(d)) if measure_out and not gn_groups else 0,
                  lib.mf_conv2d_f16x2_sync_words(C.byref(d)))
    need, slots, words = pinned
    if not measure_out or gn_groups:
        slots = 0
    yb = torch.empty((d.N, slots), dtype=torch.float32, device=x1.device) if slots else None
    partial = torch.empty((d.N, gn_parts, gn_groups, 2), dtype=torch.float64, device=x1.device) if gn_groups else None
    ws = Workspace.get(need, x1.device) if need else None
    sync = SyncWords.get(words, x1.device) if words else None
    rc = lib.mf_conv2d_f16x2(x1s.data_ptr(), _ptr(x2s), wh.data_ptr(), _ptr(bias), out.data_ptr(), b1.data_ptr(), _ptr(b2), wmax, _ptr(yb), _ptr(ws), need,
                             _ptr(sync), _ptr(partial), gn_groups, C.byref(d), stream())
    L.check(rc, "mf_conv2d_f16x2")
    if slots:   # per-(tile, wave) maxima: reduced to the bound of each sample by the first consumer that asks (bound_of), or inside the
        out._mf_slots = yb   # GroupNorm-apply pass that takes this tensor as its residual (without slots a consumer measures on demand)
        _stamp(out)
    return (out, partial) if gn_groups else out


def conv_group_ok(da: L.MfConvDesc, Ga: int, db: L.MfConvDesc, Gb: int) -> bool:
    """can the two fp16-pair convolutions share one launch (mf_conv2d_f16x2_group_ok)?"""
    return bool(L.load().mf_conv2d_f16x2_group_ok(C.byref(da), Ga, C.byref(db), Gb))


def conv2d_f16x2_group(x1: torch.Tensor, x2: Optional[torch.Tensor], a, b):
    """TWO independent fp16-pair convolutions of the same input (x1 | x2) in ONE launch (mf_conv2d_f16x2_group): a = the large one, followed by a
    GroupNorm -- dict(w_split, bias, d, gn_groups, gn_parts, pinned) -> (y, partial records) --, b = the small one whose output is measured --
    dict(w_split, bias, d, pinned) -> y with its bound slots attached.  Bit for bit what conv2d_f16x2(.., gn_groups=..) and
    conv2d_f16x2(.., measure_out=True) return."""
    _gpu(x1, x2, a["w_split"][0], b["w_split"][0], a["bias"], b["bias"])
    lib = L.load()
    dev = x1.device
    x1s, b1 = split_of(x1), bound_of(x1)
    x2s, b2 = (split_of(x2), bound_of(x2)) if x2 is not None else (None, None)
    da, db = a["d"], b["d"]
    need_a, _, words_a = a["pinned"]
    need_b, slots_b, words_b = b["pinned"]
    off_b = (need_a + 255) & ~255                      # b's hand-off region behind a's, its counters behind a's
    ws = Workspace.get(off_b + need_b, dev) if (need_a or need_b) else None
    sync = SyncWords.get(words_a + words_b, dev) if (words_a or words_b) else None
    hoa, woa = conv_out_hw(da)
    hob, wob = conv_out_hw(db)
    ya = torch.empty((da.N, hoa, woa, da.Cout), dtype=torch.float32, device=dev)
    yb = torch.empty((db.N, hob, wob, db.Cout), dtype=torch.float32, device=dev)
    G, parts = a["gn_groups"], a["gn_parts"]
    partial = torch.empty((da.N, parts, G, 2), dtype=torch.float64, device=dev)
    slots = torch.empty((db.N, slots_b), dtype=torch.float32, device=dev) if slots_b else None
    (wa, wmax_a), (wb, wmax_b) = a["w_split"], b["w_split"]
    ca = L.MfConvF16x2Call(x1s.data_ptr(), _ptr(x2s), wa.data_ptr(), _ptr(a["bias"]), ya.data_ptr(), b1.data_ptr(), _ptr(b2), wmax_a, None,
                           ws.data_ptr() if need_a else None, need_a, sync.data_ptr() if words_a else None, partial.data_ptr(), G, C.pointer(da))
    cb = L.MfConvF16x2Call(x1s.data_ptr(), _ptr(x2s), wb.data_ptr(), _ptr(b["bias"]), yb.data_ptr(), b1.data_ptr(), _ptr(b2), wmax_b, _ptr(slots),
                           ws.data_ptr() + off_b if need_b else None, need_b, sync.data_ptr() + 4 * words_a if words_b else None, None, 0, C.pointer(db))
    L.check(lib.mf_conv2d_f16x2_group(C.byref(ca), C.byref(cb), stream()), "mf_conv2d_f16x2_group")
    if slots_b:
        yb._mf_slots = slots
        _stamp(yb)
    return (ya, partial), yb


def pack_nchw_pairs(x_nchw: torch.Tensor, cp: int = 32) -> torch.Tensor:
    """NCHW fp32 [N, C, H, W] (C <= cp) -> an NHWC [N, H, W, cp] tensor that exists ONLY as fp16 pairs (channels C.. zero), scaled per sample by
    the max |x| the same launch measures (mf_pack_nchw_pairs_f32): the operand form of the network input"""
    _gpu(x_nchw)
    x = x_nchw.contiguous()
    n, c, h, w = x.shape
    out = torch.empty((n, h, w, cp), dtype=torch.float32, device=x.device)     # (storage never written: pairs only)
    outs = torch.empty((n, h, w, cp), dtype=torch.int32, device=x.device)
    ob = torch.empty((n,), dtype=torch.float32, device=x.device)
    L.check(L.load().mf_pack_nchw_pairs_f32(x.data_ptr(), outs.data_ptr(), ob.data_ptr(), n, c, h * w, cp, stream()), "mf_pack_nchw_pairs_f32")
    out._mf_split, out._mf_bound, out._mf_pairs_only = outs, ob, True
    _stamp(out)
    _audit(out, "pack_nchw_pairs (measured)")
    return out


def conv_pairs_out_ok(d: L.MfConvDesc) -> bool:
    return bool(L.load().mf_conv2d_f16x2_pairs_out_ok(C.byref(d)))


def conv2d_f16x2_pairs_out(x1: torch.Tensor, w_split, bias: Optional[torch.Tensor], d: L.MfConvDesc, l1, bias_max: float, x2: Optional[torch.Tensor] = None,
                           pinned=None) -> torch.Tensor:
    """the fp16-pair convolution writing its output as fp32 AND as fp16 pairs under a bound derived from the operands
    (mf_conv2d_f16x2_pairs_out): l1 = (largest filter L1 norm over the channels of x1, ... of x2), bias_max = max |bias|.  The result carries
    its mirror and bound like the output of gn_apply(split=True): no measuring pass, no split launch in front of its consumers."""
    wh, wmax = w_split
    _gpu(x1, x2, wh, bias)
    lib = L.load()
    x1s, b1 = split_of(x1), bound_of(x1)
    x2s, b2 = (split_of(x2), bound_of(x2)) if x2 is not None else (None, None)
    ho, wo = conv_out_hw(d)
    out = torch.empty((d.N, ho, wo, d.Cout), dtype=torch.float32, device=x1.device)
    outs = torch.empty((d.N, ho, wo, d.Cout), dtype=torch.int32, device=x1.device)
    ob = torch.empty((d.N,), dtype=torch.float32, device=x1.device)
    if pinned is None:
        pinned = (lib.mf_conv2d_workspace_bytes(C.byref(d)), 0, lib.mf_conv2d_f16x2_sync_words(C.byref(d)))
    need, _, words = pinned
    ws = Workspace.get(need, x1.device) if need else None
    sync = SyncWords.get(words, x1.device) if words else None
    rc = lib.mf_conv2d_f16x2_pairs_out(x1s.data_ptr(), _ptr(x2s), wh.data_ptr(), _ptr(bias), out.data_ptr(), outs.data_ptr(), ob.data_ptr(), b1.data_ptr(), _ptr(b2),
                                       wmax, float(l1[0]), float(l1[1]), float(bias_max), _ptr(ws), need, _ptr(sync), C.byref(d), stream())
    L.check(rc, "mf_conv2d_f16x2_pairs_out")
    out._mf_split, out._mf_bound = outs, ob
    _stamp(out)
    _audit(out, "conv pairs_out (derived: bound(x) L1(w) + max|bias|)")
    return out


# ----------------------------------------------------------------------------- Winograd F(2x2, 3x3) form of the 3x3 stride-1 convolutions
# The transform-domain mirror of an activation travels as attributes of the tensor, like the plain fp16-pair mirror: `t._mf_wino` (int32
# [16, N, T, C]: V = B^T d B as fp16 pairs) and `t._mf_wino_bound` (float [16 N]).  Producers that write it themselves set both; a consumer that
# finds neither runs the stand-alone input transform once (mf_wino_input_f16x2, from the plain pair mirror) and caches the result on the tensor.
def wino_ok(d: L.MfConvDesc) -> bool:
    return bool(L.load().mf_wino_ok(C.byref(d)))


def wino_preferred(d: L.MfConvDesc) -> bool:
    """is the Winograd form the faster one for `d` on MI355X?  (mf_wino_preferred: since ABI 240 a rule in (Cin, Cout, H W, N) that holds at any batch --
    csrc/conv_f16x2_wino.inc; the host switch is blocks.WINOGRAD / MEDFUSION_WINOGRAD = 0 never | 1 where preferred | 2 wherever mf_wino_ok)"""
    return bool(L.load().mf_wino_preferred(C.byref(d)))


def wino_gn_parts(d: L.MfConvDesc, G: int) -> int:
    return L.load().mf_wino_gn_parts(C.byref(d), G)


def wino_pack_weight(w_oihw: torch.Tensor) -> torch.Tensor:
    """OIHW 3x3 -> U = G g G^T as [16, Cout, 1, 1, Cin] fp32 (device, once at load)"""
    _gpu(w_oihw)
    w = w_oihw.contiguous()
    co, ci, kh, kw = w.shape
    assert (kh, kw) == (3, 3)
    out = torch.empty((16, co, 1, 1, ci), dtype=torch.float32, device=w.device)
    L.check(L.load().mf_wino_pack_weight_f32(w.data_ptr(), out.data_ptr(), co, ci, stream()), "mf_wino_pack_weight_f32")
    return out


def wino_input(x: torch.Tensor):
    """(V as fp16 pairs [16, N, T, C], bounds [16 N]) of the NHWC activation x, cached on x.  A tensor whose producer could have written V
    itself (conv2d_wino_gn_apply: `_mf_wino_site`) tells it so here: from its next call on that producer emits V with its output -- the
    stand-alone transform below runs once per site (the eager warm-up iteration of a sampling loop), never in the recorded iteration."""
    v = _fresh(x, "_mf_wino")
    if v is None:
        site = getattr(x, "_mf_wino_site", None)
        if site is not None:
            site[0][site[1]] = True
        xs, xb = split_of(x), bound_of(x)
        n, h, w, c = x.shape
        v = torch.empty((16, n, (h // 2) * (w // 2), c), dtype=torch.int32, device=x.device)
        vb = torch.empty((16 * n,), dtype=torch.float32, device=x.device)
        L.check(L.load().mf_wino_input_f16x2(xs.data_ptr(), xb.data_ptr(), v.data_ptr(), vb.data_ptr(), n, h, w, c, stream()), "mf_wino_input_f16x2")
        x._mf_wino, x._mf_wino_bound = v, vb
        _stamp(x)
    return v, x._mf_wino_bound


def pin_wino_plan(d: L.MfConvDesc):
    """(workspace bytes, sync words) of the Winograd form of `d`, computed once by callers that launch the same descriptor every iteration"""
    lib = L.load()
    return lib.mf_wino_workspace_bytes(C.byref(d)), lib.mf_wino_sync_words(C.byref(d))


def conv2d_wino_f16x2(x1: torch.Tensor, u_split, bias: Optional[torch.Tensor], d: L.MfConvDesc, x2: Optional[torch.Tensor] = None,
                      gn_groups: int = 0, gn_parts: int = 0, pinned=None):
    """The 3x3 stride-1 convolution `d` in its Winograd form (mf_conv2d_wino_f16x2): u_split = split_weight_f16x2(wino_pack_weight(w)).
    Returns y fp32 NHWC, or (y, partial [N, parts, G, 2]) when gn_groups > 0 (gn_parts = wino_gn_parts(d, G) > 0)."""
    uh, umax = u_split
    _gpu(x1, x2, uh, bias)
    lib = L.load()
    v1, b1 = wino_input(x1)
    v2, b2 = wino_input(x2) if x2 is not None else (None, None)
    dev = x1.device
    out = torch.empty((d.N, d.Hin, d.Win, d.Cout), dtype=torch.float32, device=dev)
    partial = torch.empty((d.N, gn_parts, gn_groups, 2), dtype=torch.float64, device=dev) if gn_groups else None
    need, words = pinned if pinned is not None else pin_wino_plan(d)
    ws = Workspace.get(need, dev)
    sync = SyncWords.get(words, dev) if words else None
    rc = lib.mf_conv2d_wino_f16x2(v1.data_ptr(), _ptr(v2), uh.data_ptr(), _ptr(bias), out.data_ptr(), b1.data_ptr(), _ptr(b2), umax, ws.data_ptr(), need,
                                  _ptr(sync), _ptr(partial), gn_groups, C.byref(d), stream())
    L.check(rc, "mf_conv2d_wino_f16x2")
    return (out, partial) if gn_groups else out


def wino_tail_ok(d: L.MfConvDesc, G: int) -> bool:
    return bool(L.load().mf_wino_tail_ok(C.byref(d), G))


def wino_group_ok(d: L.MfConvDesc, guest: L.MfConvDesc) -> bool:
    """can the fp16-pair convolution `guest` share the launch of d's component GEMM (mf_wino_group_ok)?"""
    return bool(L.load().mf_wino_group_ok(C.byref(d), C.byref(guest)))


def conv2d_wino_gn_apply(x1: torch.Tensor, u_split, bias: Optional[torch.Tensor], d: L.MfConvDesc, gamma, beta, G: int, eps: float, act: int = 1,
                         residual: Optional[torch.Tensor] = None, emb: Optional[torch.Tensor] = None, emb_stride: int = 0,
                         x2: Optional[torch.Tensor] = None, bconst: float = 0.0, out_fp32: bool = True, want_wino: bool = False, pinned=None,
                         guest=None) -> torch.Tensor:
    """Winograd conv -> GroupNorm -> Swish -> (+ residual) -> (+ emb) (-> the input transform of the next Winograd convolution) in two launches
    (mf_conv2d_wino_gn_apply_f16x2): the component GEMM and one tail.  Returns the result with its fp16-pair mirror and bound attached (out_fp32 =
    False: pairs only, like gn_apply) and, with want_wino, its transform-domain mirror as well.
    guest = dict(w_split, bias, d, pinned): conv_res of the same ResBlock (a 1x1 on the same x1 | x2) in the GEMM's grid; its measured output IS the
    residual of the tail (`residual` must be None then)."""
    uh, umax = u_split
    _gpu(x1, x2, uh, bias, gamma, beta, residual, emb)
    lib = L.load()
    v1, b1 = wino_input(x1)
    v2, b2 = wino_input(x2) if x2 is not None else (None, None)
    n, h, w, c = d.N, d.Hin, d.Win, d.Cout
    dev = x1.device
    need, words = pinned if pinned is not None else pin_wino_plan(d)
    gcall = None
    if guest is not None:
        if residual is not None:
            raise RuntimeError("conv2d_wino_gn_apply: the guest's output is the residual")
        dg = guest["d"]
        need_g, slots_g, words_g = guest["pinned"]
        off_g = (need + 255) & ~255                      # the guest's hand-off region behind the GEMM's workspace, its counters behind the GEMM's
        ws = Workspace.get(off_g + need_g, dev)
        sync = SyncWords.get(words + words_g, dev) if (words or words_g) else None
        x1s, xb1 = split_of(x1), bound_of(x1)
        x2s, xb2 = (split_of(x2), bound_of(x2)) if x2 is not None else (None, None)
        hog, wog = conv_out_hw(dg)
        residual = torch.empty((dg.N, hog, wog, dg.Cout), dtype=torch.float32, device=dev)
        gslots = torch.empty((dg.N, slots_g), dtype=torch.float32, device=dev) if slots_g else None
        wg, wgmax = guest["w_split"]
        gcall = L.MfConvF16x2Call(x1s.data_ptr(), _ptr(x2s), wg.data_ptr(), _ptr(guest["bias"]), residual.data_ptr(), xb1.data_ptr(), _ptr(xb2), wgmax, _ptr(gslots),
                                  ws.data_ptr() + off_g if need_g else None, need_g, sync.data_ptr() + 4 * words if words_g else None, None, 0, C.pointer(dg))
        if slots_g:
            residual._mf_slots = gslots
            _stamp(residual)
    else:
        ws = Workspace.get(need, dev)
        sync = SyncWords.get(words, dev) if words else None
    res_pairs = rb = rslots = eb = None
    if residual is not None:
        if pairs_only(residual):
            res_pairs, rb = residual._mf_split, residual._mf_bound
        elif _fresh(residual, "_mf_bound") is None and _fresh(residual, "_mf_slots") is not None:
            rslots = residual._mf_slots
        else:
            rb = bound_of(residual)
    if emb is not None:
        eb = _fresh(emb, "_mf_bound")
        if eb is None:
            eb = maxabs_rows(emb.contiguous())
    out = torch.empty((n, h, w, c), dtype=torch.float32, device=dev)
    outs = torch.empty((n, h, w, c), dtype=torch.int32, device=dev)
    ob = torch.empty((n,), dtype=torch.float32, device=dev)
    ov = torch.empty((16, n, (h // 2) * (w // 2), c), dtype=torch.int32, device=dev) if want_wino else None
    ovb = torch.empty((16 * n,), dtype=torch.float32, device=dev) if want_wino else None
    t = L.MfWinoTail(_ptr(gamma), _ptr(beta), None if res_pairs is not None else _ptr(residual), _ptr(res_pairs), _ptr(rb), _ptr(rslots), _ptr(emb), _ptr(eb),
                     out.data_ptr() if out_fp32 else None, outs.data_ptr(), ob.data_ptr(), _ptr(ov), _ptr(ovb), int(emb_stride),
                     0 if rslots is None else rslots.shape[1], int(act), float(bconst), float(eps))
    rc = lib.mf_conv2d_wino_gn_apply_f16x2(v1.data_ptr(), _ptr(v2), uh.data_ptr(), _ptr(bias), b1.data_ptr(), _ptr(b2), umax, ws.data_ptr(), need, _ptr(sync), G,
                                           C.byref(t), None if gcall is None else C.byref(gcall), C.byref(d), stream())
    L.check(rc, "mf_conv2d_wino_gn_apply_f16x2")
    out._mf_split, out._mf_bound = outs, ob
    if want_wino:
        out._mf_wino, out._mf_wino_bound = ov, ovb
    _stamp(out)
    _audit(out, "Winograd tail (derived: bconst + residual + embedding)")
    if not out_fp32:
        out._mf_pairs_only = True
    return out


# ----------------------------------------------------------------------------- the Winograd form on the EXACT arithmetics (round 6, ABI 250)
def wino_f32_ok(d: L.MfConvDesc, G: int) -> bool:
    """can the 3x3 `d` (precision 0 = fp32 MFMA or 3 = exact bf16 triplets) with the G-group GroupNorm behind it run as fp32 transforms + component GEMMs
    on the exact arithmetic's own kernel + the tail?  (mf_wino_f32_ok)"""
    return bool(L.load().mf_wino_f32_ok(C.byref(d), G))


def wino_input_f32(x: torch.Tensor) -> torch.Tensor:
    """V = B^T d B of the fp32 NHWC activation x as fp32 [16, N, T, C], cached on x; a tensor whose producer could have written V itself
    (conv2d_wino_gn_apply_f32: `_mf_wino_site_f32`) tells it so here, like wino_input does for the fp16-pair form"""
    v = _fresh(x, "_mf_wino_f32")
    if v is None:
        _need_f32(x)
        site = getattr(x, "_mf_wino_site_f32", None)
        if site is not None:
            site[0][site[1]] = True
        n, h, w, c = x.shape
        v = torch.empty((16, n, (h // 2) * (w // 2), c), dtype=torch.float32, device=x.device)
        L.check(L.load().mf_wino_input_f32(x.data_ptr(), v.data_ptr(), n, h, w, c, stream()), "mf_wino_input_f32")
        x._mf_wino_f32 = v
        _stamp(x)
    return v


def conv2d_wino_gn_apply_f32(x1: torch.Tensor, u_packed: torch.Tensor, bias: Optional[torch.Tensor], d: L.MfConvDesc, gamma, beta, G: int, eps: float,
                             act: int = 1, residual: Optional[torch.Tensor] = None, emb: Optional[torch.Tensor] = None, emb_stride: int = 0,
                             x2: Optional[torch.Tensor] = None, want_wino: bool = False) -> torch.Tensor:
    """conv3x3 -> GroupNorm -> Swish -> + residual -> + emb on the Winograd form of an EXACT arithmetic (d.precision 0 or 3): the 16 component GEMMs run on
    that arithmetic's own implicit-GEMM kernel (mf_conv2d_f32 on the upsample = 3 descriptor; u_packed = U = G g G^T [16, Cout, 1, 1, Cin], pre-split for
    precision 3), the transforms and the tail are fp32.  Returns y fp32 NHWC; want_wino: it also carries V of y for the next Winograd convolution."""
    _gpu(x1, x2, u_packed, bias, gamma, beta, residual, emb)
    lib = L.load()
    n, h, w, c1 = x1.shape
    c2 = 0 if x2 is None else x2.shape[-1]
    co, t = d.Cout, (h // 2) * (w // 2)
    v1 = wino_input_f32(x1)
    v2 = wino_input_f32(x2) if x2 is not None else None
    g = make_conv_desc(16 * n, 1, t, c1, c2, co, 1, 1, 0, 3, precision=d.precision)
    m = conv2d(v1.view(16 * n, 1, t, c1), u_packed, None, g, x2=None if v2 is None else v2.view(16 * n, 1, t, c2))
    if residual is not None:
        _need_f32(residual)
    out = torch.empty((n, h, w, co), dtype=torch.float32, device=x1.device)
    ov = torch.empty((16, n, t, co), dtype=torch.float32, device=x1.device) if want_wino else None
    L.check(lib.mf_wino_tail_f32(m.data_ptr(), _ptr(bias), _ptr(gamma), _ptr(beta), _ptr(residual), _ptr(emb), emb_stride, out.data_ptr(), _ptr(ov),
                                 n, h, w, co, G, act, eps, stream()), "mf_wino_tail_f32")
    if want_wino:
        out._mf_wino_f32 = ov
        _stamp(out)
    return out


def conv_fuse_words(d: L.MfConvDesc, G: int) -> int:
    """rendezvous words mf_conv2d_f16x2_gn_apply needs for `d` followed by a G-group GroupNorm; 0: this convolution cannot apply it itself
    (capability; whether blocks.py uses the form is Rendezvous.disabled / FUSE_MIN_HW)"""
    return L.load().mf_conv2d_f16x2_fuse_words(C.byref(d), G)


def conv2d_f16x2_gn_apply(x1: torch.Tensor, w_split, bias: Optional[torch.Tensor], d: L.MfConvDesc, gamma, beta, G: int, eps: float, parts: int,
                          words: int, act: int = 1, residual: Optional[torch.Tensor] = None, emb: Optional[torch.Tensor] = None, emb_stride: int = 0,
                          x2: Optional[torch.Tensor] = None, bconst: float = 0.0, out_fp32: bool = True, pinned=None) -> torch.Tensor:
    """conv -> GroupNorm -> Swish -> (+ residual) -> (+ emb) in ONE launch (mf_conv2d_f16x2_gn_apply): what conv2d_f16x2(gn_groups=G) followed
    by gn_apply(GnPartials, split=True) computes, bit for bit.  words = conv_fuse_words(d, G) > 0, parts = conv_gn_parts(d, G) > 0.
    Returns the result with its fp16-pair mirror and bound attached (out_fp32=False: pairs only, like gn_apply)."""
    wh, wmax = w_split
    _gpu(x1, x2, wh, bias, gamma, beta, residual, emb)
    lib = L.load()
    x1s, b1 = split_of(x1), bound_of(x1)
    x2s, b2 = (split_of(x2), bound_of(x2)) if x2 is not None else (None, None)
    ho, wo = conv_out_hw(d)
    n, c = d.N, d.Cout
    dev = x1.device
    res_pairs = rb = rslots = eb = None
    if residual is not None:
        if pairs_only(residual):
            res_pairs, rb = residual._mf_split, residual._mf_bound
        elif _fresh(residual, "_mf_bound") is None and _fresh(residual, "_mf_slots") is not None:
            rslots = residual._mf_slots
        else:
            rb = bound_of(residual)
    if emb is not None:
        eb = _fresh(emb, "_mf_bound")
        if eb is None:
            eb = maxabs_rows(emb.contiguous())
    out = torch.empty((n, ho, wo, c), dtype=torch.float32, device=dev)
    outs = torch.empty((n, ho, wo, c), dtype=torch.int32, device=dev)
    ob = torch.empty((n,), dtype=torch.float32, device=dev)
    partial = torch.empty((n, parts, G, 2), dtype=torch.float64, device=dev)
    if pinned is None:
        pinned = (lib.mf_conv2d_workspace_bytes(C.byref(d)), 0, lib.mf_conv2d_f16x2_sync_words(C.byref(d)))
    need, _, swords = pinned
    ws = Workspace.get(need, dev) if need else None
    sync = SyncWords.get(swords, dev) if swords else None
    rv = Rendezvous.get(words, dev)
    f = L.MfGnFuse(_ptr(gamma), _ptr(beta), None if res_pairs is not None else _ptr(residual), _ptr(res_pairs), _ptr(rb), _ptr(rslots), _ptr(emb), _ptr(eb),
                   out.data_ptr() if out_fp32 else None, outs.data_ptr(), ob.data_ptr(), rv.data_ptr() + 16, rv.data_ptr(), int(emb_stride),
                   0 if rslots is None else rslots.shape[1], int(act), float(bconst), float(eps))
    rc = lib.mf_conv2d_f16x2_gn_apply(x1s.data_ptr(), _ptr(x2s), wh.data_ptr(), _ptr(bias), b1.data_ptr(), _ptr(b2), wmax, _ptr(ws), need, _ptr(sync),
                                      partial.data_ptr(), G, C.byref(f), C.byref(d), stream())
    L.check(rc, "mf_conv2d_f16x2_gn_apply")
    out._mf_split, out._mf_bound = outs, ob
    _stamp(out)
    _audit(out, "conv + GroupNorm tail in one launch (derived)")
    if not out_fp32:
        out._mf_pairs_only = True
    return out


def make_conv_desc(N, Hin, Win, C1, C2, Cout, k, stride, pad, upsample=0, in_layout=L.LAYOUT_NHWC, out_layout=L.LAYOUT_NHWC,
                   tile_hint=0, splitk_hint=0, precision=0) -> L.MfConvDesc:
    return L.MfConvDesc(N, Hin, Win, C1, C2, Cout, k, k, stride, pad, upsample, in_layout, out_layout, tile_hint, splitk_hint, precision)


def convert_conv_weight_bf16(w_packed: torch.Tensor) -> torch.Tensor:
    """packed fp32 weights -> bf16 (round to nearest even) for the opt-in MF_CONV_BF16 mode: opaque int16 tensor [rows, K]"""
    _gpu(w_packed)
    w = w_packed.contiguous()
    k = w.shape[-1] * w.shape[-2] * w.shape[-3]
    rows = w.numel() // k
    out = torch.empty((rows, k), dtype=torch.int16, device=w.device)
    L.check(L.load().mf_convert_conv_weight_bf16(w.data_ptr(), out.data_ptr(), rows, k, stream()), "mf_convert_conv_weight_bf16")
    return out


def conv_out_hw(d: L.MfConvDesc):
    up = 1 if d.upsample else 0
    he, we = d.Hin << up, d.Win << up
    return (he + 2 * d.pad - d.KH) // d.stride + 1, (we + 2 * d.pad - d.KW) // d.stride + 1


def conv2d(x1: torch.Tensor, w_packed: torch.Tensor, bias: Optional[torch.Tensor], d: L.MfConvDesc, x2: Optional[torch.Tensor] = None,
           out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """y = conv(x1 ++ x2) + bias per descriptor `d`.  Output NHWC [N,Ho,Wo,Cout] or NCHW [N,Cout,Ho,Wo]."""
    _gpu(x1, x2, w_packed, bias)
    _need_f32(x1, x2)
    lib = L.load()
    ho, wo = conv_out_hw(d)
    if out is None:
        shape = (d.N, d.Cout, ho, wo) if d.out_layout == L.LAYOUT_NCHW else (d.N, ho, wo, d.Cout)
        out = torch.empty(shape, dtype=torch.float32, device=x1.device)
    else:
        drop_split(out)
    need = lib.mf_conv2d_workspace_bytes(C.byref(d))
    ws = Workspace.get(need, x1.device) if need else None
    rc = lib.mf_conv2d_f32(x1.data_ptr(), _ptr(x2), w_packed.data_ptr(), _ptr(bias), out.data_ptr(), _ptr(ws), need, C.byref(d), stream())
    L.check(rc, "mf_conv2d_f32")
    return out


def conv_gn_parts(d: L.MfConvDesc, G: int) -> int:
    """How many per-sample partial GroupNorm records this convolution emits itself (0: it cannot)."""
    return L.load().mf_conv2d_gn_parts(C.byref(d), G)


def conv2d_gn(x1: torch.Tensor, w_packed: torch.Tensor, bias: Optional[torch.Tensor], d: L.MfConvDesc, G: int, parts: int,
              x2: Optional[torch.Tensor] = None):
    """Convolution + the partial statistics of the GroupNorm that follows (conv epilogue or split-K reducer) -> (y NHWC, partial)."""
    _gpu(x1, x2, w_packed, bias)
    _need_f32(x1, x2)
    lib = L.load()
    ho, wo = conv_out_hw(d)
    out = torch.empty((d.N, ho, wo, d.Cout), dtype=torch.float32, device=x1.device)
    partial = torch.empty((d.N, parts, G, 2), dtype=torch.float64, device=x1.device)
    need = lib.mf_conv2d_workspace_bytes(C.byref(d))
    ws = Workspace.get(need, x1.device) if need else None
    rc = lib.mf_conv2d_gn_f32(x1.data_ptr(), _ptr(x2), w_packed.data_ptr(), _ptr(bias), out.data_ptr(), _ptr(ws), need, partial.data_ptr(), G, C.byref(d),
                              stream())
    L.check(rc, "mf_conv2d_gn_f32")
    return out, partial


def gn_stats_partial(x: torch.Tensor, G: int):
    """x NHWC -> (partial [N, parts, G, 2] float64 {sum, sumsq}, parts)"""
    if x.dtype != torch.float32:
        raise RuntimeError("gn_stats_partial: fp32 only")
    _gpu(x)
    _need_f32(x)
    n, h, w, c = x.shape
    lib = L.load()
    parts = lib.mf_gn_partial_parts(h * w)
    partial = torch.empty((n, parts, G, 2), dtype=torch.float64, device=x.device)
    L.check(lib.mf_gn_stats_partial_f32(x.data_ptr(), partial.data_ptr(), n, h * w, c, G, stream()), "mf_gn_stats_partial_f32")
    return partial, parts


def gn_finalize(partial: torch.Tensor, parts: int, HW: int, C: int, G: int, eps: float = 1e-5) -> torch.Tensor:
    """partial [N, parts, G, 2] -> stats [N, G, 2] = (mean, rstd)"""
    _gpu(partial)
    n = partial.shape[0]
    stats = torch.empty((n, G, 2), dtype=torch.float32, device=partial.device)
    L.check(L.load().mf_gn_finalize_f32(partial.data_ptr(), parts, stats.data_ptr(), n, HW, C, G, eps, stream()), "mf_gn_finalize_f32")
    return stats


def gn_stats(x: torch.Tensor, G: int, eps: float = 1e-5) -> torch.Tensor:
    """x NHWC [N,H,W,C] -> stats [N,G,2] = (mean, rstd)."""
    _gpu(x)
    _need_f32(x)
    n, h, w, c = x.shape
    lib = L.load()
    stats = torch.empty((n, G, 2), dtype=torch.float32, device=x.device)
    need = lib.mf_gn_stats_workspace_bytes(n, h * w, c, G)
    ws = Workspace.get(need, x.device)
    L.check(lib.mf_gn_stats_f32(x.data_ptr(), stats.data_ptr(), ws.data_ptr(), need, n, h * w, c, G, eps, stream()), "mf_gn_stats_f32")
    return stats


class GnPartials:
    """the partial GroupNorm sums a convolution left ([N, parts, G, 2] fp64) handed to gn_apply un-finalised: the apply pass reduces them itself"""
    __slots__ = ("records", "parts", "eps")

    def __init__(self, records, parts, eps):
        self.records, self.parts, self.eps = records, parts, eps


def gn_apply(x: torch.Tensor, stats, gamma, beta, G: int, act: int = 1, residual: Optional[torch.Tensor] = None,
             emb: Optional[torch.Tensor] = None, emb_stride: int = 0, out: Optional[torch.Tensor] = None, split: bool = False,
             bconst: float = 0.0, out_fp32: bool = True) -> torch.Tensor:
    """stats: [N, G, 2] mean / rstd (gn_stats / gn_finalize), a GnPartials (mf_gn_apply_from_partials_f32: no finalize launch) or None.
    split=True: also emit the fp16-pair mirror of the result (operand of a following MF_CONV_FP32_F16X2 convolution), scaled per sample
    by the bound the pass derives: bconst (>= max |act(gn(x) gamma + beta)|, from the caller; the bound of x when nothing is normalised)
    + the bounds of the residual and of the embedding rows.
    out_fp32=False (a REQUEST, honoured on the from-partials pass with split=True): the result is wanted as fp16 pairs only -- every consumer
    is an fp16-pair convolution or the residual input of another such pass -- and its fp32 form is not written (12 instead of 16 bytes per
    element); the returned tensor says so (pairs_only).  A `residual` that is itself pairs-only is read from its pairs."""
    part = stats if isinstance(stats, GnPartials) else None
    if part is not None:
        stats = None
    _gpu(x, stats, gamma, beta, residual, emb)
    _need_f32(x)
    n, h, w, c = x.shape
    split = split and c % 8 == 0
    res_pairs = None
    if pairs_only(residual):
        if not (split and part is not None):
            _need_f32(residual)
        res_pairs = residual._mf_split
    want_pairs_only = (not out_fp32) and split and part is not None
    xb = rb = eb = ob = outs = rslots = None
    if split:  # (before `out` may alias x or the residual: their bounds describe the values this pass READS)
        xb = bound_of(x) if (stats is None and part is None) else None
        if residual is not None:
            if part is not None and _fresh(residual, "_mf_bound") is None and _fresh(residual, "_mf_slots") is not None:
                rslots = residual._mf_slots
            else:
                rb = bound_of(residual)
        if emb is not None:
            eb = _fresh(emb, "_mf_bound")
            if eb is None:
                eb = maxabs_rows(emb.contiguous())
    if out is None:
        out = torch.empty_like(x)
    else:
        drop_split(out)
    if split:
        outs = torch.empty(out.shape, dtype=torch.int32, device=x.device)
        ob = torch.empty((n,), dtype=torch.float32, device=x.device)
    if part is not None:
        rc = L.load().mf_gn_apply_from_partials_pairs_f32(x.data_ptr(), part.records.data_ptr(), part.parts, float(part.eps), _ptr(gamma), _ptr(beta),
                                                          None if res_pairs is not None else _ptr(residual), _ptr(res_pairs), _ptr(emb), emb_stride,
                                                          None if want_pairs_only else out.data_ptr(), _ptr(outs), _ptr(rb), _ptr(rslots),
                                                          0 if rslots is None else rslots.shape[1], _ptr(eb), float(bconst), _ptr(ob), n, h * w, c, G,
                                                          act, stream())
        L.check(rc, "mf_gn_apply_from_partials_pairs_f32")
    else:
        rc = L.load().mf_gn_apply_split_f32(x.data_ptr(), _ptr(stats), _ptr(gamma), _ptr(beta), _ptr(residual), _ptr(emb), emb_stride, out.data_ptr(),
                                            _ptr(outs), _ptr(xb), _ptr(rb), _ptr(eb), float(bconst), _ptr(ob), n, h * w, c, G, act, stream())
        L.check(rc, "mf_gn_apply_split_f32")
    if split:
        out._mf_split, out._mf_bound = outs, ob
        _stamp(out)
        _audit(out, "GroupNorm apply (derived: bconst + residual + embedding)" if (stats is not None or part is not None) else "apply without norm (bound(x) + residual + embedding)")
        if want_pairs_only:
            out._mf_pairs_only = True
    return out


def linear(x: torch.Tensor, w: torch.Tensor, bias: Optional[torch.Tensor], act_in: bool = False, act_out: bool = False,
           out: Optional[torch.Tensor] = None, accumulate: bool = False) -> torch.Tensor:
    """x [B, In] (row stride may exceed In), w [Out, In] -> [B, Out]."""
    _gpu(x, w, bias)
    b, inn = x.shape
    o = w.shape[0]
    assert w.shape[1] == inn and x.stride(1) == 1 and w.is_contiguous()
    if out is None:
        out = torch.empty((b, o), dtype=torch.float32, device=x.device)
    else:
        drop_split(out)
    rc = L.load().mf_linear_f32(x.data_ptr(), x.stride(0), w.data_ptr(), _ptr(bias), out.data_ptr(), out.stride(0), b, inn, o, int(act_in),
                                int(act_out), int(accumulate), stream())
    L.check(rc, "mf_linear_f32")
    return out


def sinusoidal(t: torch.Tensor, dim: int, max_period: float = 10000.0, shift: float = 1.0, flip: bool = False,
               freqs: Optional[torch.Tensor] = None) -> torch.Tensor:
    _gpu(t, freqs)
    t = t.to(torch.float32).contiguous()
    out = torch.empty((t.shape[0], dim), dtype=torch.float32, device=t.device)
    L.check(L.load().mf_sinusoidal_f32(t.data_ptr(), _ptr(freqs), out.data_ptr(), t.shape[0], dim, max_period, shift, int(flip), stream()),
            "mf_sinusoidal_f32")
    return out


def learned_sinusoidal(t: torch.Tensor, weights: torch.Tensor, emb_dim: int) -> torch.Tensor:
    """LearnedSinusoidalPosEmb.forward (time_embedder.py:42-49): [B] -> [B, 1 + 2 (emb_dim // 2) + (emb_dim & 1)]"""
    _gpu(t, weights)
    t = t.to(torch.float32).contiguous()
    w = weights.detach().contiguous()
    out = torch.empty((t.shape[0], 1 + 2 * (emb_dim // 2) + (emb_dim & 1)), dtype=torch.float32, device=t.device)
    L.check(L.load().mf_learned_sinusoidal_f32(t.data_ptr(), w.data_ptr(), out.data_ptr(), t.shape[0], emb_dim, stream()), "mf_learned_sinusoidal_f32")
    return out


def embedding_add(table: torch.Tensor, idx: torch.Tensor, io: torch.Tensor) -> torch.Tensor:
    _gpu(table, idx, io)
    idx = idx.to(torch.int64).contiguous()
    drop_split(io)
    L.check(L.load().mf_embedding_add_f32(table.data_ptr(), idx.data_ptr(), io.data_ptr(), io.shape[0], io.shape[1], table.shape[0], stream()),
            "mf_embedding_add_f32")
    return io


def philox_normal(out: torch.Tensor, seed: int, draw: int, sample_offset: int = 0, step_dev: Optional[torch.Tensor] = None,
                  draw_stride: int = 0) -> torch.Tensor:
    """Fill out [B, ...] with N(0,1): draw index = draw + draw_stride * (*step_dev or 0)."""
    _gpu(out, step_dev)
    b = out.shape[0]
    per = out.numel() // b
    drop_split(out)
    rc = L.load().mf_philox_normal_f32(out.data_ptr(), seed & 0xFFFFFFFFFFFFFFFF, draw, draw_stride, _ptr(step_dev), 0, sample_offset, b, per, stream())
    L.check(rc, "mf_philox_normal_f32")
    return out


def rows_axpby(x: torch.Tensor, a: Optional[torch.Tensor] = None, y: Optional[torch.Tensor] = None, c: Optional[torch.Tensor] = None,
               d: Optional[torch.Tensor] = None, clamp: Optional[tuple] = None, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """out[b] = clamp((a[b]*x[b] + c[b]*y[b]) / d[b]); a, c, d are [B] fp32 device vectors (or None)."""
    _gpu(x, a, y, c, d)
    x = x.contiguous()
    if y is not None:
        y = y.contiguous()
    b = x.shape[0]
    if out is None:
        out = torch.empty_like(x)
    else:
        drop_split(out)
    lo, hi = clamp if clamp is not None else (0.0, 0.0)
    rc = L.load().mf_rows_axpby_f32(x.data_ptr(), _ptr(y), _ptr(a), _ptr(c), _ptr(d), out.data_ptr(), b, x.numel() // b, int(clamp is not None),
                                    float(lo), float(hi), stream())
    L.check(rc, "mf_rows_axpby_f32")
    return out


def image_to_uint8(x_nchw: torch.Tensor, normalize_each: bool = False) -> torch.Tensor:
    """[N,C,H,W] float in ~[-1,1] -> [N,H,W,C] uint8 on the device (mode 0: sample_dataset.py; mode 1: sample.py + save_image)."""
    _gpu(x_nchw)
    x = x_nchw.contiguous()
    n, c, h, w = x.shape
    out = torch.empty((n, h, w, c), dtype=torch.uint8, device=x.device)
    ws = torch.empty((n, 2), dtype=torch.float32, device=x.device) if normalize_each else None
    L.check(L.load().mf_image_egress_u8(x.data_ptr(), out.data_ptr(), _ptr(ws), n, c, h, w, int(normalize_each), stream()), "mf_image_egress_u8")
    return out


def sched_step(args: L.MfSchedArgs, outputs=()) -> None:
    """`outputs`: the tensors the step writes (x_t, x0, ...): the call goes by raw pointers, their fp16-pair mirrors (if any) are stale after it"""
    for t in outputs:
        drop_split(t)
    L.check(L.load().mf_sched_step_f32(C.byref(args), stream()), "mf_sched_step_f32")


def sched_step_philox(args: L.MfSchedArgs, seed: int, draw_base: int, draw_stride: int, sample_offset: int, B: int, counter: torch.Tensor, outputs=()) -> None:
    """the tail of a denoise iteration in one launch (mf_sched_step_philox_f32): both noise draws in registers, the scheduler step, the
    step counter += 1.  counter: int32 [2] on the device = (step, ticket word)."""
    _gpu(counter)
    for t in outputs:
        drop_split(t)
    L.check(L.load().mf_sched_step_philox_f32(C.byref(args), seed & 0xFFFFFFFFFFFFFFFF, draw_base, draw_stride, sample_offset, B, counter.data_ptr(),
                                              counter.data_ptr() + 4, stream()), "mf_sched_step_philox_f32")


def gather_step_rows_multi(tables, step, cols: torch.Tensor):
    """gather_step_rows for up to three [S, NCOL, L_i] tables that share `cols` and the step, in ONE launch -> list of [B, L_i]"""
    _gpu(cols, *tables)
    n = len(tables)
    assert 1 <= n <= 3 and all(t.is_contiguous() and t.shape[:2] == tables[0].shape[:2] for t in tables) and cols.dtype == torch.int64 and cols.is_contiguous()
    B, ncol = cols.shape[0], tables[0].shape[1]
    outs = [torch.empty((B, t.shape[2]), dtype=torch.float32, device=t.device) for t in tables]
    tp = (L.c_fp * n)(*[t.data_ptr() for t in tables])
    op = (L.c_fp * n)(*[o.data_ptr() for o in outs])
    rl = (C.c_int64 * n)(*[t.shape[2] for t in tables])
    dev_step = isinstance(step, torch.Tensor)
    rc = L.load().mf_gather_step_rows3_f32(tp, rl, op, n, cols.data_ptr(), step.data_ptr() if dev_step else None, 0 if dev_step else int(step), ncol, B, stream())
    L.check(rc, "mf_gather_step_rows3_f32")
    return outs


def gather_step_rows(table: torch.Tensor, step, cols: torch.Tensor) -> torch.Tensor:
    """table [S, NCOL, L] fp32, cols [B] int64 -> [B, L] = table[step, cols[b]]; `step`: a host int or a device int32 counter (graph replay)"""
    _gpu(table, cols)
    s_, ncol, ln = table.shape
    assert table.is_contiguous() and cols.dtype == torch.int64 and cols.is_contiguous()
    out = torch.empty((cols.shape[0], ln), dtype=torch.float32, device=table.device)
    dev_step = isinstance(step, torch.Tensor)
    rc = L.load().mf_gather_step_rows_f32(table.data_ptr(), cols.data_ptr(), step.data_ptr() if dev_step else None, 0 if dev_step else int(step), ncol, ln,
                                          out.data_ptr(), cols.shape[0], stream())
    L.check(rc, "mf_gather_step_rows_f32")
    return out


def broadcast_from_table(table: torch.Tensor, step_dev: torch.Tensor, out: torch.Tensor) -> torch.Tensor:
    _gpu(table, step_dev, out)
    drop_split(out)
    L.check(L.load().mf_broadcast_from_table_f32(table.data_ptr(), step_dev.data_ptr(), 0, out.data_ptr(), out.numel(), stream()), "mf_broadcast_from_table_f32")
    return out


def counter_add(counter: torch.Tensor, inc: int = 1) -> None:
    _gpu(counter)
    L.check(L.load().mf_counter_add_i32(counter.data_ptr(), inc, stream()), "mf_counter_add_i32")


def attention(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, heads: int, scale: float) -> torch.Tensor:
    """q [B,Nq,C], k/v [B,Nk,C] -> [B,Nq,C]."""
    _gpu(q, k, v)
    b, nq, c = q.shape
    nk = k.shape[1]
    out = torch.empty_like(q)
    L.check(L.load().mf_attention_f32(q.data_ptr(), k.data_ptr(), v.data_ptr(), out.data_ptr(), b, heads, nq, nk, c // heads, scale, stream()),
            "mf_attention_f32")
    return out


def layernorm(x: torch.Tensor, gamma, beta, eps: float = 1e-5) -> torch.Tensor:
    _gpu(x, gamma, beta)
    c = x.shape[-1]
    out = torch.empty_like(x)
    L.check(L.load().mf_layernorm_f32(x.data_ptr(), gamma.data_ptr(), beta.data_ptr(), out.data_ptr(), x.numel() // c, c, eps, stream()), "mf_layernorm_f32")
    return out


def geglu(h: torch.Tensor) -> torch.Tensor:
    _gpu(h)
    c = h.shape[-1] // 2
    out = torch.empty((*h.shape[:-1], c), dtype=torch.float32, device=h.device)
    L.check(L.load().mf_geglu_f32(h.data_ptr(), out.data_ptr(), h.numel() // (2 * c), c, stream()), "mf_geglu_f32")
    return out


def add(a: torch.Tensor, b: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    _gpu(a, b)
    _need_f32(a, b)
    if out is None:
        out = torch.empty_like(a)
    else:
        drop_split(out)
    L.check(L.load().mf_add_f32(a.data_ptr(), b.data_ptr(), out.data_ptr(), a.numel(), stream()), "mf_add_f32")
    return out


def nchw_to_nhwc(x: torch.Tensor) -> torch.Tensor:
    _gpu(x)
    n, c, h, w = x.shape
    x = x.contiguous()
    out = torch.empty((n, h, w, c), dtype=torch.float32, device=x.device)
    L.check(L.load().mf_nchw_to_nhwc_f32(x.data_ptr(), out.data_ptr(), n, c, h, w, stream()), "mf_nchw_to_nhwc_f32")
    return out


def nhwc_to_nchw(x: torch.Tensor) -> torch.Tensor:
    _gpu(x)
    _need_f32(x)
    n, h, w, c = x.shape
    out = torch.empty((n, c, h, w), dtype=torch.float32, device=x.device)
    L.check(L.load().mf_nhwc_to_nchw_f32(x.data_ptr(), out.data_ptr(), n, c, h, w, stream()), "mf_nhwc_to_nchw_f32")
    return out


def avgpool2d(x: torch.Tensor, k: int, stride: int, pad: int) -> torch.Tensor:
    """nn.AvgPool2d(k, stride, pad) on NHWC (count_include_pad like torch's default)"""
    _gpu(x)
    _need_f32(x)
    n, h, w, c = x.shape
    ho, wo = (h + 2 * pad - k) // stride + 1, (w + 2 * pad - k) // stride + 1
    out = torch.empty((n, ho, wo, c), dtype=torch.float32, device=x.device)
    L.check(L.load().mf_avgpool2d_nhwc_f32(x.contiguous().data_ptr(), out.data_ptr(), n, h, w, c, k, stride, pad, stream()), "mf_avgpool2d_nhwc_f32")
    return out


def upsample_nearest2x(x: torch.Tensor) -> torch.Tensor:
    """F.interpolate(scale 2, nearest-exact) on NHWC"""
    _gpu(x)
    _need_f32(x)
    n, h, w, c = x.shape
    out = torch.empty((n, 2 * h, 2 * w, c), dtype=torch.float32, device=x.device)
    L.check(L.load().mf_upsample_nearest2x_nhwc_f32(x.contiguous().data_ptr(), out.data_ptr(), n, h, w, c, stream()), "mf_upsample_nearest2x_nhwc_f32")
    return out


def pixel_unshuffle2_add(x: torch.Tensor, y: torch.Tensor) -> torch.Tensor:
    """y [N, H/2, W/2, 4C] += nn.PixelUnshuffle(2)(x [N, H, W, C]) on NHWC, in place (BasicDown(use_res=True))"""
    _gpu(x, y)
    _need_f32(x, y)
    n, h, w, c = x.shape
    if y.shape != (n, h // 2, w // 2, 4 * c) or h % 2 or w % 2:
        raise RuntimeError(f"pixel_unshuffle2_add: x {tuple(x.shape)} does not unshuffle onto y {tuple(y.shape)}")
    drop_split(y)
    L.check(L.load().mf_pixel_unshuffle2_add_nhwc_f32(x.contiguous().data_ptr(), y.data_ptr(), n, h, w, c, stream()), "mf_pixel_unshuffle2_add_nhwc_f32")
    return y


def pixel_shuffle2_add(x: torch.Tensor, y: torch.Tensor) -> torch.Tensor:
    """y [N, 2H, 2W, C/4] += nn.PixelShuffle(2)(x [N, H, W, C]) on NHWC, in place (BasicUp(use_res=True))"""
    _gpu(x, y)
    _need_f32(x, y)
    n, h, w, c = x.shape
    if c % 4 or y.shape != (n, 2 * h, 2 * w, c // 4):
        raise RuntimeError(f"pixel_shuffle2_add: x {tuple(x.shape)} does not shuffle onto y {tuple(y.shape)}")
    drop_split(y)
    L.check(L.load().mf_pixel_shuffle2_add_nhwc_f32(x.contiguous().data_ptr(), y.data_ptr(), n, h, w, c, stream()), "mf_pixel_shuffle2_add_nhwc_f32")
    return y


def diag_gaussian_sample(moments_nchw: torch.Tensor, noise: torch.Tensor) -> torch.Tensor:
    _gpu(moments_nchw, noise)
    n, c2, h, w = moments_nchw.shape
    z = torch.empty((n, c2 // 2, h, w), dtype=torch.float32, device=moments_nchw.device)
    L.check(L.load().mf_diag_gaussian_sample_f32(moments_nchw.data_ptr(), noise.data_ptr(), z.data_ptr(), n, c2 // 2, h * w, stream()),
            "mf_diag_gaussian_sample_f32")
    return z


def diag_gaussian_kl(moments_nchw: torch.Tensor) -> torch.Tensor:
    """the KL term of DiagonalGaussianDistribution (latent_embedders.py:29-31) as a 0-dim device tensor"""
    _gpu(moments_nchw)
    n, c2, h, w = moments_nchw.shape
    kl = torch.empty((1,), dtype=torch.float32, device=moments_nchw.device)
    L.check(L.load().mf_diag_gaussian_kl_f32(moments_nchw.contiguous().data_ptr(), kl.data_ptr(), n, c2 // 2, h * w, stream()), "mf_diag_gaussian_kl_f32")
    return kl.reshape(())


# ----------------------------------------------------------------------------- launch timing
_PROF_ACTIVE = [False]


def prof_active() -> bool:
    """launch timing is on (the command-list loop steps aside: replayed launches would not be timed)"""
    return _PROF_ACTIVE[0]


class prof:
    """with prof() as p: ...; p.table() -> {family: (ms, launches, algorithmic flops, bytes, executed flops)}"""

    def __enter__(self):
        lib = L.load()
        lib.mf_prof_reset()
        lib.mf_prof_enable(1)
        _PROF_ACTIVE[0] = True
        return self

    def __exit__(self, *exc):
        torch.cuda.synchronize()
        L.load().mf_prof_enable(0)
        _PROF_ACTIVE[0] = False
        return False

    @staticmethod
    def table() -> dict:
        lib = L.load()
        out = {}
        for i, name in enumerate(L.FAMILIES):
            ms, n, fl, by, ex = C.c_double(), C.c_int64(), C.c_double(), C.c_double(), C.c_double()
            L.check(lib.mf_prof_query2(i, C.byref(ms), C.byref(n), C.byref(fl), C.byref(by), C.byref(ex)), "mf_prof_query2")
            if n.value:
                out[name] = (ms.value, n.value, fl.value, by.value, ex.value)
        return out


def prof_rows(family: str):
    """the records of `family` (a name of lib.FAMILIES) since prof() started, grouped by kernel instantiation:
    [{kernel (as rocprofv3 prints it), variant, launches, ms, flops, bytes, exec_flops}]"""
    lib = L.load()
    fam = L.FAMILIES.index(family)
    rows = (L.MfProfRow * 64)()
    n = lib.mf_prof_rows(fam, rows, 64)
    if n < 0:
        L.check(n, "mf_prof_rows")
    out = []
    buf = C.create_string_buffer(256)
    for r in rows[:n]:
        lib.mf_prof_tag_name(fam, r.tag, buf, 256)
        out.append(dict(kernel=buf.value.decode(), tag=r.tag, variant=r.variant, launches=r.launches, ms=r.ms, flops=r.flops, bytes=r.bytes, exec_flops=r.exec_flops))
    return out


def mfma_sustained_tflops(device, data: str = "random", iters: int = 4000, reps: int = 3) -> float:
    """TFLOP/s the fp16 matrix pipe sustains on `device` with nothing but MFMAs in flight (mf_mfma_rate_probe_f16), operands "random"
    (N(0,1) as fp16) or "zeros": the measured ceiling bench.py sets next to the nominal peak."""
    wgs = torch.cuda.get_device_properties(device).multi_processor_count
    n = wgs * 512 * 8 * 8
    ops = (torch.randn((n,), device=device, generator=torch.Generator(device=device).manual_seed(1)) if data == "random"
           else torch.zeros((n,), device=device)).to(torch.float16)
    out = torch.empty((wgs * 512,), dtype=torch.float32, device=device)
    fl = C.c_double(0.0)
    lib = L.load()
    best = 0.0
    for r in range(reps + 1):   # first launch: warm-up of the same length
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        L.check(lib.mf_mfma_rate_probe_f16(ops.data_ptr(), out.data_ptr(), wgs, iters, C.byref(fl), stream()), "mf_mfma_rate_probe_f16")
        e1.record()
        e1.synchronize()
        if r:
            best = max(best, fl.value / (e0.elapsed_time(e1) * 1e-3) / 1e12)
    return best
