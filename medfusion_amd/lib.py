"""ctypes binding of libmedfusion_hip.so (see include/medfusion_hip.h).

The product path has NO CPU fallback: if the library is missing it is built with hipcc, and if
that fails (or a tensor is not on a ROCm device) a RuntimeError is raised.
"""
from __future__ import annotations

import ctypes as C
import os
from pathlib import Path

HERE = Path(__file__).resolve().parent
LIB_PATH = HERE / "libmedfusion_hip.so"

c_fp = C.c_void_p  # device pointers travel as integers (tensor.data_ptr())


class MfConvDesc(C.Structure):
    _fields_ = [(n, C.c_int32) for n in (
        "N", "Hin", "Win", "C1", "C2", "Cout", "KH", "KW", "stride", "pad", "upsample", "in_layout", "out_layout",
        "tile_hint", "splitk_hint", "precision")]


class MfSchedStep(C.Structure):
    _fields_ = [("sqrt_recip_ac", C.c_float), ("sqrt_recipm1_ac", C.c_float), ("coef1", C.c_float), ("coef2", C.c_float),
                ("std_fixed", C.c_float), ("log_var_min", C.c_float), ("log_var_max", C.c_float), ("ddim_sqrt_an", C.c_float),
                ("ddim_c", C.c_float), ("ddim_sigma", C.c_float), ("t", C.c_int32), ("mode", C.c_int32)]


class MfSchedArgs(C.Structure):
    _fields_ = [("x_t", c_fp), ("pred", c_fp), ("pred_uncond", c_fp), ("pred_var", c_fp), ("noise_post", c_fp), ("noise_ddim", c_fp),
                ("noise_step_stride", C.c_int64), ("x_t_out", c_fp), ("x0_out", c_fp), ("xT_out", c_fp), ("table", c_fp),
                ("step_dev", c_fp), ("step", C.c_int32), ("objective", C.c_int32), ("clip_x0", C.c_int32),
                ("guidance_scale", C.c_float), ("n", C.c_int64)]


class MfGnFuse(C.Structure):
    _fields_ = [("gamma", c_fp), ("beta", c_fp), ("residual", c_fp), ("residual_pairs", c_fp), ("res_bound", c_fp), ("res_bound_slots", c_fp),
                ("emb", c_fp), ("emb_bound", c_fp), ("out", c_fp), ("out_split", c_fp), ("out_bound", c_fp), ("rendezvous", c_fp), ("error_flag", c_fp),
                ("emb_stride", C.c_int64), ("res_nslots", C.c_int32), ("act", C.c_int32), ("bconst", C.c_float), ("eps", C.c_float)]


class MfWinoTail(C.Structure):
    _fields_ = [("gamma", c_fp), ("beta", c_fp), ("residual", c_fp), ("residual_pairs", c_fp), ("res_bound", c_fp), ("res_bound_slots", c_fp),
                ("emb", c_fp), ("emb_bound", c_fp), ("out", c_fp), ("out_split", c_fp), ("out_bound", c_fp), ("out_wino", c_fp), ("wino_bound", c_fp),
                ("emb_stride", C.c_int64), ("res_nslots", C.c_int32), ("act", C.c_int32), ("bconst", C.c_float), ("eps", C.c_float)]


class MfProfRow(C.Structure):
    _fields_ = [("tag", C.c_int32), ("variant", C.c_int32), ("launches", C.c_int64), ("ms", C.c_double), ("flops", C.c_double), ("bytes", C.c_double),
                ("exec_flops", C.c_double)]


class MfConvF16x2Call(C.Structure):
    """the arguments of one mf_conv2d_f16x2 call (mf_conv2d_f16x2_group takes two)"""
    _fields_ = [("x1s", c_fp), ("x2s", c_fp), ("ws", c_fp), ("bias", c_fp), ("y", c_fp), ("x1_bound", c_fp), ("x2_bound", c_fp), ("w_bound", C.c_float),
                ("y_bound", c_fp), ("workspace", c_fp), ("workspace_bytes", C.c_size_t), ("sync", c_fp), ("gn_partial", c_fp), ("G", C.c_int32),
                ("d", C.POINTER(MfConvDesc))]


LAYOUT_NHWC, LAYOUT_NCHW = 0, 1
FAMILIES = ("conv_igemm", "conv_direct", "splitk_reduce", "gn_stats", "gn_apply", "linear", "sched", "noise", "attention", "misc", "conv_gn_fused", "wino_xform")

_I, _I64, _F, _SZ, _U64 = C.c_int, C.c_int64, C.c_float, C.c_size_t, C.c_uint64
_SIGS = {
    "mf_version": (C.c_int, []),
    "mf_last_error": (C.c_char_p, []),
    "mf_pack_conv_weight_f32": (_I, [c_fp, c_fp, _I, _I, _I, _I, c_fp]),
    "mf_pack_upconv_weight_f32": (_I, [c_fp, c_fp, _I, _I, c_fp]),
    "mf_conv2d_subpixel_ok": (_I, [C.POINTER(MfConvDesc)]),
    "mf_conv2d_is_igemm": (_I, [C.POINTER(MfConvDesc)]),
    "mf_split_conv_weight_bf16x3": (_I, [c_fp, c_fp, C.c_long, _I, c_fp]),
    "mf_convert_conv_weight_bf16": (_I, [c_fp, c_fp, C.c_long, _I, c_fp]),
    "mf_conv2d_workspace_bytes": (_SZ, [C.POINTER(MfConvDesc)]),
    "mf_conv2d_f16x2_ok": (_I, [C.POINTER(MfConvDesc)]),
    "mf_split_f16x2": (_I, [c_fp, c_fp, c_fp, _I, _I64, c_fp]),
    "mf_split_f16x2_slots": (_I, [c_fp, c_fp, c_fp, _I, c_fp, _I, _I64, c_fp]),
    "mf_conv2d_f16x2": (_I, [c_fp, c_fp, c_fp, c_fp, c_fp, c_fp, c_fp, _F, c_fp, c_fp, _SZ, c_fp, c_fp, _I, C.POINTER(MfConvDesc), c_fp]),
    "mf_conv2d_f16x2_sync_words": (_I, [C.POINTER(MfConvDesc)]),
    "mf_pack_nchw_pairs_f32": (_I, [c_fp, c_fp, c_fp, _I, _I, _I, _I, c_fp]),
    "mf_conv2d_f16x2_pairs_out_ok": (_I, [C.POINTER(MfConvDesc)]),
    "mf_conv2d_f16x2_pairs_out": (_I, [c_fp, c_fp, c_fp, c_fp, c_fp, c_fp, c_fp, c_fp, c_fp, _F, _F, _F, _F, c_fp, _SZ, c_fp, C.POINTER(MfConvDesc), c_fp]),
    "mf_conv2d_f16x2_fuse_words": (_I, [C.POINTER(MfConvDesc), _I]),
    "mf_conv2d_f16x2_gn_apply": (_I, [c_fp, c_fp, c_fp, c_fp, c_fp, c_fp, _F, c_fp, _SZ, c_fp, c_fp, _I, C.POINTER(MfGnFuse), C.POINTER(MfConvDesc), c_fp]),
    "mf_conv2d_f16x2_group_ok": (_I, [C.POINTER(MfConvDesc), _I, C.POINTER(MfConvDesc), _I]),
    "mf_conv2d_f16x2_group": (_I, [C.POINTER(MfConvF16x2Call), C.POINTER(MfConvF16x2Call), c_fp]),
    "mf_wino_ok": (_I, [C.POINTER(MfConvDesc)]),
    "mf_wino_preferred": (_I, [C.POINTER(MfConvDesc)]),
    "mf_wino_in_table": (_I, [C.POINTER(MfConvDesc)]),
    "mf_wino_pack_weight_f32": (_I, [c_fp, c_fp, _I, _I, c_fp]),
    "mf_wino_input_f16x2": (_I, [c_fp, c_fp, c_fp, c_fp, _I, _I, _I, _I, c_fp]),
    "mf_wino_workspace_bytes": (_SZ, [C.POINTER(MfConvDesc)]),
    "mf_wino_sync_words": (_I, [C.POINTER(MfConvDesc)]),
    "mf_wino_gn_parts": (_I, [C.POINTER(MfConvDesc), _I]),
    "mf_wino_plan_query": (_I, [C.POINTER(MfConvDesc), C.POINTER(C.c_int32), C.POINTER(C.c_int32)]),
    "mf_wino_f32_ok": (_I, [C.POINTER(MfConvDesc), _I]),
    "mf_wino_input_f32": (_I, [c_fp, c_fp, _I, _I, _I, _I, c_fp]),
    "mf_wino_tail_f32": (_I, [c_fp, c_fp, c_fp, c_fp, c_fp, c_fp, _I64, c_fp, c_fp, _I, _I, _I, _I, _I, _I, _F, c_fp]),
    "mf_conv2d_wino_f16x2": (_I, [c_fp, c_fp, c_fp, c_fp, c_fp, c_fp, c_fp, _F, c_fp, _SZ, c_fp, c_fp, _I, C.POINTER(MfConvDesc), c_fp]),
    "mf_wino_tail_ok": (_I, [C.POINTER(MfConvDesc), _I]),
    "mf_wino_group_ok": (_I, [C.POINTER(MfConvDesc), C.POINTER(MfConvDesc)]),
    "mf_conv2d_wino_gn_apply_f16x2": (_I, [c_fp, c_fp, c_fp, c_fp, c_fp, c_fp, _F, c_fp, _SZ, c_fp, _I, C.POINTER(MfWinoTail), C.POINTER(MfConvF16x2Call),
                                           C.POINTER(MfConvDesc), c_fp]),
    "mf_maxabs_rows_slots": (_I, [_I64]),
    "mf_maxabs_rows_f32": (_I, [c_fp, c_fp, c_fp, _I, _I64, c_fp]),
    "mf_bound_finalize_f32": (_I, [c_fp, c_fp, _I, _I, c_fp]),
    "mf_conv2d_f16x2_bound_slots": (_I, [C.POINTER(MfConvDesc)]),
    "mf_conv2d_plan_override": (_I, [C.POINTER(MfConvDesc), _I, _I]),
    "mf_conv2d_plan_query": (_I, [C.POINTER(MfConvDesc), C.POINTER(C.c_int32), C.POINTER(C.c_int32)]),
    "mf_gn_apply_split_f32": (_I, [c_fp, c_fp, c_fp, c_fp, c_fp, c_fp, _I64, c_fp, c_fp, c_fp, c_fp, c_fp, _F, c_fp, _I, _I, _I, _I, _I, c_fp]),
    "mf_gn_apply_from_partials_f32": (_I, [c_fp, c_fp, _I, _F, c_fp, c_fp, c_fp, c_fp, _I64, c_fp, c_fp, c_fp, c_fp, _I, c_fp, _F, c_fp, _I, _I, _I, _I, _I, c_fp]),
    "mf_gn_apply_from_partials_pairs_f32": (_I, [c_fp, c_fp, _I, _F, c_fp, c_fp, c_fp, c_fp, c_fp, _I64, c_fp, c_fp, c_fp, c_fp, _I, c_fp, _F, c_fp, _I, _I, _I, _I, _I, c_fp]),
    "mf_conv2d_f32": (_I, [c_fp, c_fp, c_fp, c_fp, c_fp, c_fp, _SZ, C.POINTER(MfConvDesc), c_fp]),
    "mf_conv2d_gn_parts": (_I, [C.POINTER(MfConvDesc), _I]),
    "mf_conv2d_gn_f32": (_I, [c_fp, c_fp, c_fp, c_fp, c_fp, c_fp, _SZ, c_fp, _I, C.POINTER(MfConvDesc), c_fp]),
    "mf_gn_partial_parts": (_I, [_I]),
    "mf_gn_finalize_f32": (_I, [c_fp, _I, c_fp, _I, _I, _I, _I, _F, c_fp]),
    "mf_gn_stats_partial_f32": (_I, [c_fp, c_fp, _I, _I, _I, _I, c_fp]),
    "mf_gn_stats_workspace_bytes": (_SZ, [_I, _I, _I, _I]),
    "mf_gn_stats_f32": (_I, [c_fp, c_fp, c_fp, _SZ, _I, _I, _I, _I, _F, c_fp]),
    "mf_gn_apply_f32": (_I, [c_fp, c_fp, c_fp, c_fp, c_fp, c_fp, _I64, c_fp, _I, _I, _I, _I, _I, c_fp]),
    "mf_linear_f32": (_I, [c_fp, _I64, c_fp, c_fp, c_fp, _I64, _I, _I, _I, _I, _I, _I, c_fp]),
    "mf_sinusoidal_f32": (_I, [c_fp, c_fp, c_fp, _I, _I, _F, _F, _I, c_fp]),
    "mf_learned_sinusoidal_f32": (_I, [c_fp, c_fp, c_fp, _I, _I, c_fp]),
    "mf_embedding_add_f32": (_I, [c_fp, c_fp, c_fp, _I, _I, _I, c_fp]),
    "mf_sched_step_f32": (_I, [C.POINTER(MfSchedArgs), c_fp]),
    "mf_sched_step_philox_f32": (_I, [C.POINTER(MfSchedArgs), _U64, C.c_int32, C.c_int32, _I64, _I, c_fp, c_fp, c_fp]),
    "mf_gather_step_rows3_f32": (_I, [C.POINTER(c_fp), C.POINTER(_I64), C.POINTER(c_fp), _I, c_fp, c_fp, C.c_int32, _I, _I, c_fp]),
    "mf_broadcast_from_table_f32": (_I, [c_fp, c_fp, C.c_int32, c_fp, _I, c_fp]),
    "mf_gather_step_rows_f32": (_I, [c_fp, c_fp, c_fp, C.c_int32, _I, _I64, c_fp, _I, c_fp]),
    "mf_rows_axpby_f32": (_I, [c_fp, c_fp, c_fp, c_fp, c_fp, c_fp, _I, _I64, _I, _F, _F, c_fp]),
    "mf_image_egress_u8": (_I, [c_fp, c_fp, c_fp, _I, _I, _I, _I, _I, c_fp]),
    "mf_avgpool2d_nhwc_f32": (_I, [c_fp, c_fp, _I, _I, _I, _I, _I, _I, _I, c_fp]),
    "mf_upsample_nearest2x_nhwc_f32": (_I, [c_fp, c_fp, _I, _I, _I, _I, c_fp]),
    "mf_pixel_unshuffle2_add_nhwc_f32": (_I, [c_fp, c_fp, _I, _I, _I, _I, c_fp]),
    "mf_pixel_shuffle2_add_nhwc_f32": (_I, [c_fp, c_fp, _I, _I, _I, _I, c_fp]),
    "mf_counter_add_i32": (_I, [c_fp, C.c_int32, c_fp]),
    "mf_philox_normal_f32": (_I, [c_fp, _U64, C.c_int32, C.c_int32, c_fp, C.c_int32, _I64, _I, _I64, c_fp]),
    "mf_attention_f32": (_I, [c_fp, c_fp, c_fp, c_fp, _I, _I, _I, _I, _I, _F, c_fp]),
    "mf_layernorm_f32": (_I, [c_fp, c_fp, c_fp, c_fp, _I64, _I, _F, c_fp]),
    "mf_geglu_f32": (_I, [c_fp, c_fp, _I64, _I, c_fp]),
    "mf_add_f32": (_I, [c_fp, c_fp, c_fp, _I64, c_fp]),
    "mf_nchw_to_nhwc_f32": (_I, [c_fp, c_fp, _I, _I, _I, _I, c_fp]),
    "mf_nhwc_to_nchw_f32": (_I, [c_fp, c_fp, _I, _I, _I, _I, c_fp]),
    "mf_diag_gaussian_sample_f32": (_I, [c_fp, c_fp, c_fp, _I, _I, _I, c_fp]),
    "mf_diag_gaussian_kl_f32": (_I, [c_fp, c_fp, _I, _I, _I, c_fp]),
    "mf_prof_enable": (_I, [_I]),
    "mf_prof_reset": (_I, []),
    "mf_prof_query": (_I, [_I, C.POINTER(C.c_double), C.POINTER(_I64), C.POINTER(C.c_double), C.POINTER(C.c_double)]),
    "mf_prof_query2": (_I, [_I, C.POINTER(C.c_double), C.POINTER(_I64), C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_double)]),
    "mf_prof_family_name": (C.c_char_p, [_I]),
    "mf_prof_rows": (_I, [_I, C.POINTER(MfProfRow), _I]),
    "mf_prof_tag_name": (_I, [_I, _I, C.c_char_p, _I]),
    "mf_mfma_rate_probe_f16": (_I, [c_fp, c_fp, _I, _I, C.POINTER(C.c_double), c_fp]),
    "mf_cmdlist_begin": (_I, []),
    "mf_cmdlist_end": (_I, [C.POINTER(C.c_void_p)]),
    "mf_cmdlist_count": (_I, [c_fp]),
    "mf_cmdlist_replay": (_I, [c_fp, _I, c_fp]),
    "mf_cmdlist_free": (_I, [c_fp]),
}

_lib = None


def load(build_if_missing: bool = True) -> C.CDLL:
    """Load (building first if needed) the HIP library.  Raises RuntimeError -- never falls back.
    MEDFUSION_LIB=<path to a .so> loads that file as it is (diagnostic twins of medfusion_amd.build.build_variant)."""
    global _lib
    if _lib is not None:
        return _lib
    path = LIB_PATH
    override = os.environ.get("MEDFUSION_LIB")
    if override:
        path = Path(override)
        if not path.exists():
            raise RuntimeError(f"MEDFUSION_LIB={override}: no such file")
    elif build_if_missing:
        try:
            import fcntl

            from . import build as _build

            # one builder at a time, and nobody decides "fresh" while another process is between its compile and its rename: the ranks
            # of a multi-GPU launch import the package together (the lock is taken BEFORE the staleness check)
            with open(str(LIB_PATH) + ".lock", "w") as lock:
                fcntl.flock(lock, fcntl.LOCK_EX)
                if _build.needs_build():
                    _build.build(verbose=bool(os.environ.get("MEDFUSION_VERBOSE_BUILD")))
        except Exception as e:  # hipcc missing etc. -- fine if a prebuilt .so travelled with the tree
            if not LIB_PATH.exists():
                raise RuntimeError(f"libmedfusion_hip.so is missing and could not be built: {e}") from e
    if not path.exists():
        raise RuntimeError(f"{path} not found: build it with `python -m medfusion_amd.build` (hipcc, gfx950)")
    lib = C.CDLL(str(path))
    for name, (res, args) in _SIGS.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is not exported
        fn.restype, fn.argtypes = res, args
    _lib = lib
    return lib


def last_error() -> str:
    return load().mf_last_error().decode(errors="replace")


def check(rc: int, what: str) -> None:
    if rc != 0:
        raise RuntimeError(f"{what} failed (code {rc}): {last_error()}")


def exported_symbols():
    return sorted(_SIGS)
