"""ctypes binding of libcfgpp_hip.so (the C ABI declared in include/cfgpp.h).

There is NO fallback: if the shared library is missing or a call fails, a
``CfgppError`` is raised.  The product path never routes through ``oracle/``.
"""
from __future__ import annotations

import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("CFGPP_LIB") or os.path.join(HERE, "libcfgpp_hip.so")   # CFGPP_LIB: A/B a second build (development only)


class CfgppError(RuntimeError):
    pass


class UNetConfigC(C.Structure):
    _fields_ = [
        ("in_channels", C.c_int), ("out_channels", C.c_int),
        ("num_levels", C.c_int),
        ("block_out_channels", C.c_int * 4),
        ("layers_per_block", C.c_int),
        ("level_has_attn", C.c_int * 4),
        ("transformer_depth", C.c_int * 4),
        ("num_heads", C.c_int * 4),
        ("cross_attention_dim", C.c_int),
        ("addition_embed", C.c_int),
        ("addition_time_embed_dim", C.c_int),
        ("addition_pooled_dim", C.c_int),
        ("norm_groups", C.c_int),
        ("sample_h", C.c_int), ("sample_w", C.c_int),
        ("max_rows", C.c_int),
    ]


_P, _I, _F, _L = C.c_void_p, C.c_int, C.c_float, C.c_long

# name -> (restype, argtypes); every symbol include/cfgpp.h declares (the drop-in boundary)
PROTOTYPES = {
    "cfgpp_last_error": (C.c_char_p, []),
    "cfgpp_build_id": (C.c_char_p, []),
    "cfgpp_step_ddim": (_I, [_P, _P, _P, _P, _I, _F, _F, _F, _F, _F, _I, _I, _L, _P]),
    "cfgpp_step_ddim_h": (_I, [_P, _P, _P, _P, _F, _F, _F, _F, _F, _I, _I, _L, _P]),
    "cfgpp_kdiff_input": (_I, [_P, _P, _F, _I, _L, _P]),
    "cfgpp_step_kdiff": (_I, [_P, _P, _P, _P, _P, C.POINTER(C.c_float), _I, _I, _I, _I, _L, _P]),
    "cfgpp_kdiff_denoise": (_I, [_P, _P, _P, _F, _F, _P, _P, _L, _P]),
    "cfgpp_lincomb": (_I, [_P, _P, _P, _P, _F, _F, _I, _L, _P]),
    "cfgpp_unet_create": (_P, [C.POINTER(UNetConfigC), _I]),
    "cfgpp_unet_destroy": (None, [_P]),
    "cfgpp_unet_load_tensor": (_I, [_P, C.c_char_p, _P, _I, C.POINTER(C.c_long), _I]),
    "cfgpp_unet_missing": (_I, [_P]),
    "cfgpp_unet_finalize": (_I, [_P]),
    "cfgpp_unet_set_context": (_I, [_P, _P, _I, _I, _P, _P, _I, _P]),
    "cfgpp_unet_forward": (_I, [_P, _P, _I, _I, _F, _P, _I, _P]),
    "cfgpp_sample_graph_ddim": (_I, [_P, _P, _P, _I, _I, _P, _P, _P, _I, C.POINTER(C.c_float), _I, _F, _I, _I, _P]),
    "cfgpp_unet_profile": (_I, [_P, _P, _I, _I, _F, _P, _I, _P, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_int), C.c_char_p, _L]),
    "cfgpp_unet_tuning": (_I, [_P, _I, C.POINTER(C.c_int), _I, _I]),
    "cfgpp_unet_flops": (C.c_double, [_P, _I]),
    "cfgpp_unet_device_bytes": (C.c_double, [_P]),
    "cfgpp_vae_create": (_P, [_I, _I, _I, _F, _I]),
    "cfgpp_vae_destroy": (None, [_P]),
    "cfgpp_vae_load_tensor": (_I, [_P, C.c_char_p, _P, _I, C.POINTER(C.c_long), _I]),
    "cfgpp_vae_finalize": (_I, [_P]),
    "cfgpp_vae_decode": (_I, [_P, _P, _P, _I, _P]),
    "cfgpp_vae_decode_image": (_I, [_P, _P, _P, _I, _P]),
    "cfgpp_vae_encode": (_I, [_P, _P, _P, _P, _P, _I, _P]),
    "cfgpp_vae_profile": (_I, [_P, _P, _P, _I, _P, C.c_char_p, C.c_long]),
    "cfgpp_vae_flops": (C.c_double, [_P, _I]),
    "cfgpp_text_create": (_P, [_I, _I, _I, _I, _I, _I, _I, _I, _I]),
    "cfgpp_text_destroy": (None, [_P]),
    "cfgpp_text_load_tensor": (_I, [_P, C.c_char_p, _P, _I, C.POINTER(C.c_long), _I]),
    "cfgpp_text_finalize": (_I, [_P]),
    "cfgpp_text_encode": (_I, [_P, C.POINTER(C.c_int), C.POINTER(C.c_int), _I, _I, _P, _P, _P]),
    "cfgpp_text_device_bytes": (C.c_double, [_P]),
    "cfgpp_vae_encode_flops": (C.c_double, [_P, _I]),
    "cfgpp_vae_device_bytes": (C.c_double, [_P]),
}

# test hooks and development switches (cfgpp_amd/csrc/cfgpp_debug.h): same library, not part of the boundary
DEBUG_PROTOTYPES = {
    "cfgpp_op_softmax_rows": (_I, [_P, _L, _I, _P]),
    "cfgpp_op_conv_in_ex": (_I, [_P, _I, _P, _P, _P, _I, _I, _I, _I, _I, _I, _P, _P, _F, _P]),
    "cfgpp_op_groupnorm": (_I, [_P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _F, _I, _I, _P]),
    "cfgpp_op_groupnorm_pre": (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _F, _I, _I, _P]),
    "cfgpp_op_igemm_set_gstat": (None, [_P]),
    "cfgpp_op_igemm_gstat_written": (_I, []),
    "cfgpp_groupnorm_set_prestats": (None, [_I]),
    "cfgpp_groupnorm_prestats_enabled": (_I, []),
    "cfgpp_op_layernorm": (_I, [_P, _P, _P, _P, _L, _I, _F, _P]),
    "cfgpp_op_attention_prepare_vt": (_I, [_P, _I, _I, _I, _P]),
    "cfgpp_op_attention": (_I, [_P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _P]),
    "cfgpp_op_conv_in": (_I, [_P, _I, _P, _P, _P, _I, _I, _I, _I, _I, _I, _P]),
    "cfgpp_op_vae_posterior": (_I, [_P, _P, _P, _P, _P, _P, _I, _I, _F, _P]),
    "cfgpp_op_conv_out": (_I, [_P, _P, _I, _P, _P, _I, _I, _I, _I, _I, _P]),
    "cfgpp_op_conv_out_ex": (_I, [_P, _P, _I, _P, _P, _I, _I, _I, _I, _I, _F, _F, _I, _P]),
    "cfgpp_conv_out_set_tiled": (None, [_I]),
    "cfgpp_op_sinusoid": (_I, [_P, _F, _P, _I, _I, _I, _I, _P]),
    "cfgpp_op_skinny_gemm": (_I, [_P, _I, _P, _P, _P, _I, _P, _I, _I, _I, _I, _I, _I, _P]),
    "cfgpp_op_f16_to_f32_rows": (_I, [_P, _P, _I, _I, _I, _I, _P]),
    "cfgpp_op_igemm": (_I, [_P, _P, _I, _I, _I, _I, _I, _I, _P, _I, _I, _P, _P, _I, _P, _I, _I, _P, _I, _I, _I, _P]),
    "cfgpp_op_igemm_heads": (_I, [_P, _I, _P, _I, _I, _P, _I, _P, _P, _P, _I, _I, _I, _I, _I, _I, _P]),
    "cfgpp_igemm_force_config": (None, [_I]),
    "cfgpp_igemm_set_staging": (None, [_I]),
    "cfgpp_igemm_set_staged_epilogue": (None, [_I]),
    "cfgpp_igemm_set_big_tiles": (None, [_I]),
    "cfgpp_igemm_set_tail_split": (None, [_I]),
    "cfgpp_igemm_set_autotune": (None, [_I]),
    "cfgpp_igemm_set_blocked_walk": (None, [_I]),
    "cfgpp_igemm_walk_plan_probe": (None, [_I, _I, _I, _I, _I, _I, _I, C.POINTER(C.c_int)]),
    "cfgpp_igemm_set_n_major": (None, [_I]),
    "cfgpp_igemm_force_split": (None, [_I]),
    "cfgpp_igemm_set_split_tile": (None, [_I]),
    "cfgpp_igemm_set_mf16": (None, [_I]),
    "cfgpp_igemm_set_mf16_rounds": (None, [_I]),
    "cfgpp_igemm_set_mf16_heads": (None, [_I]),
    "cfgpp_igemm_set_mf16_linear": (None, [_I]),
    "cfgpp_igemm_set_big_split": (None, [_I]),
    "cfgpp_igemm_set_tune_mask": (None, [C.c_uint]),
    "cfgpp_igemm_tuner_state": (C.c_uint, []),
    "cfgpp_igemm_timeline": (None, [_P, _L, _I]),
    "cfgpp_igemm_timeline_info": (None, [C.POINTER(C.c_int)]),
    "cfgpp_groupnorm_set_mode": (None, [_I]),
    "cfgpp_layernorm_set_rows_per_wave": (None, [_I]),
    "cfgpp_attention_set_dma": (None, [_I]),
    "cfgpp_attention_set_stagger": (None, [_I]),
    "cfgpp_attention_set_cross": (None, [_I]),
}

_lib = None


def load():
    """dlopen the library and attach prototypes.  Raises CfgppError when absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise CfgppError(
            f"{LIB_PATH} not found: build it with `python -m cfgpp_amd.build` "
            "(__graft_entry__.build()).  The HIP path has no CPU fallback.")
    # torch first: it brings its own copy of the HIP runtime; loaded after libcfgpp_hip.so (which would pull in the system one)
    # the process ends up with two runtimes and the engine cannot use torch's device pointers / streams
    import torch  # noqa: F401
    try:
        lib = C.CDLL(LIB_PATH)
    except OSError as e:  # pragma: no cover
        raise CfgppError(f"cannot load {LIB_PATH}: {e}") from e
    for name, (res, args) in list(PROTOTYPES.items()) + list(DEBUG_PROTOTYPES.items()):
        try:
            fn = getattr(lib, name)
        except AttributeError as e:
            raise CfgppError(f"{LIB_PATH} does not export {name}") from e
        fn.restype = res
        fn.argtypes = args
    if os.environ.get("CFGPP_AUTOTUNE", "1") == "0":      # e.g. under rocprofv3 --pmc: no timing passes
        lib.cfgpp_igemm_set_autotune(0)
    _lib = lib
    return lib


def build_id() -> str:
    """provenance string of the loaded binary (include/cfgpp.h: cfgpp_build_id)"""
    return load().cfgpp_build_id().decode("utf-8", "replace")


def last_error() -> str:
    return load().cfgpp_last_error().decode("utf-8", "replace")


def check(rc: int, what: str):
    if rc != 0:
        raise CfgppError(f"{what} failed (rc={rc}): {last_error()}")
