"""ctypes binding of include/simpb_hip.h. The library is mandatory: there is no CPU or PyTorch
fallback behind these entry points, and a missing .so is an error, not a silent downgrade."""
import ctypes
import os

from .build import LIB

_lib = None

_P = ctypes.c_void_p
_I = ctypes.c_int
_F = ctypes.c_float
_LL = ctypes.c_longlong

SIGNATURES = {
    "simpb_abi_version": ([], _I),
    "simpb_last_error": ([], ctypes.c_char_p),
    "simpb_timing_enable": ([_I], _I),
    "simpb_timing_read": ([_I, _P, _I], _I),
    "simpb_timing_reset": ([], None),
    "simpb_deformable_aggregation_forward": ([_P] * 6 + [_I] * 8 + [_P], _I),
    "simpb_dfa_fused_forward": ([_P, _P, _I] + [_P] * 11 + [_I] * 9 + [_P], _I),
    "simpb_deformable_aggregation_backward": ([_P] * 9 + [_I] * 8 + [_P], _I),
    "simpb_ms_deform_attn_grouped_forward": ([_P] * 7 + [_I] * 8 + [_P], _I),
    "simpb_msda_linear_forward": ([_P, _I, _P, _I, _P, _P, _P, _I, _P, _I, _P, _P] + [_I] * 8 + [_P], _I),
    "simpb_ms_deform_attn_grouped_backward": ([_P] * 10 + [_I] * 8 + [_P], _I),
    "simpb_linear_f32": ([_P] * 4 + [_I] * 4 + [_P], _I),
    "simpb_linear_f16x3": ([_P] * 5 + [_I] * 3 + [_P], _I),
    "simpb_gemm_f32": ([_P, _P], _I),
    "simpb_layernorm_f32": ([_P, _I, _P, _I, _I, _P, _I, _I, _P, _P, _I, _P, _P], _I),
    "simpb_bias_act_nhwc_f16": ([_P, _P, _P, ctypes.c_longlong, _I, _I, _P], _I),
    "simpb_bias_relu_maxpool_nhwc_f16": ([_P, _P, _P, _I, _I, _I, _I, _P], _I),
    "simpb_conv1x1_nhwc_f16": ([_P] * 5 + [_I] * 8 + [_P, _I, _P], _I),
    "simpb_conv3x3_nhwc_f16": ([_P, _P, _P, _I, _I, _P, _P, _P] + [_I] * 8 + [_P], _I),
    "simpb_conv3x3_group_tokens_f16": ([_I, _P, _P, _I, _P, _P, _P, _P, _I, _P, _P, _I, _I, _I, _P], _I),
    "simpb_stem_conv7x7_pool_f16": ([_P] * 4 + [_I] * 4 + [_P], _I),
    "simpb_image_to_nhwc4_f16": ([_P, _P] + [ctypes.c_longlong] * 4 + [_I] * 4 + [_P], _I),
    "simpb_linear_f16in_split": ([_P] * 5 + [_I] * 3 + [_P], _I),
    "simpb_format_tokens": ([_P, _P, _P, _P, _I, _I, _I, _I, _P], _I),
    "simpb_attention_f32": ([_P] * 6 + [_I] * 9 + [_F, _P], _I),
    "simpb_attention_f32_split": ([_P] * 6 + [_I] * 9 + [_F, _P], _I),
    "simpb_attention_split_halfs": ([_P] * 6 + [_I] * 9 + [_P], _I),
    "simpb_mlp_chain_forward": ([_P, _P], _I),
    "simpb_bank_get": ([_P] * 6 + [_I] * 2 + [_F] * 2 + [_P], _I),
    "simpb_bank_update": ([_P] * 10 + [_I] * 5 + [_P], _I),
    "simpb_bank_update_rank": ([_P, _P] + [_I] * 4 + [_P], _I),
    "simpb_bank_update_merge": ([_P] * 13 + [_I, _P] + [_I] * 5 + [_P], _I),
    "simpb_bank_cache": ([_P] * 10 + [_I] * 6 + [_F, _I, _F, _P, _I, _P, _P], _I),
    "simpb_bank_cache_streams": ([_P] * 10 + [_I] * 6 + [_F, _I, _F, _P, _I, _P, _P, _P], _I),
    "simpb_decode3d_record": ([_P] * 6 + [_I] * 4 + [_P], _I),
    "simpb_decode2d_record": ([_P] * 6 + [_I] * 4 + [_F] * 4 + [_P], _I),
    "simpb_record2d_compact": ([_P, _LL, _P, _LL, _I, _I, _I, _P], _I),
    "simpb_decode2d_record_ragged": ([_P] * 7 + [_I] * 5 + [_F] * 4 + [_P], _I),
    "simpb_topk_rows": ([_P, _P, _P, _I, _I, _I, _P], _I),
    "simpb_rowdot_sigmoid": ([_P, _P, _I, _P, _P, _I, _I, _P, _P], _I),
    "simpb_anchor_projection": ([_P] * 4 + [_I] * 2 + [_P], _I),
    "simpb_msda_prep": ([_P, _P, _P, _I, _P, _I, _P, _I, _I, _I, _I, _P, _P], _I),
    "simpb_dfa_points": ([_P] * 7 + [_I] * 5 + [_P], _I),
    "simpb_dfa_weights": ([_P] * 3 + [_I] * 6 + [_P], _I),
    "simpb_alloc_project": ([_P] * 5 + [_I] * 3 + [_F] * 5 + [_P], _I),
    "simpb_alloc_compact": ([_P] * 3 + [_I] * 3 + [_P], _I),
    "simpb_alloc_group_start": ([_P] * 3 + [_I] * 3 + [_P], _I),
    "simpb_alloc_scatter": ([_P] * 12 + [_I] * 4 + [_F] * 2 + [_P], _I),
    "simpb_alloc_static": ([_P] * 15 + [_I] * 4 + [_F] * 5 + [_P], _I),
    "simpb_alloc_ragged": ([_P] * 15 + [_I] * 4 + [_F] * 5 + [_P], _I),
    "simpb_gather_rows": ([_P] * 3 + [_I] * 4 + [_P], _I),
    "simpb_aggregate_2d_to_3d": ([_P] * 8 + [_I] * 5 + [_P], _I),
    "simpb_aggregate_2d_to_3d_alpha": ([_P] * 9 + [_I, _I, _P, _P] + [_I] * 5 + [_P], _I),
}

ERRORS = {1: "SIMPB_EINVAL (bad pointer/size/layout)", 2: "SIMPB_ELAUNCH (kernel launch failed)"}


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB):
            raise RuntimeError(
                f"{LIB} is missing: build it with `python -m simpb_amd.build` (hipcc --offload-arch=gfx950). "
                "simpb_amd has no fallback path for its HIP kernels.")
        # torch first: it brings its own HIP runtime (libamdhip64), and the library must bind to
        # that one. Loaded the other way round the process ends up with two runtimes and the
        # kernels launch on one that never saw the device ("no ROCm-capable device is detected").
        import torch  # noqa: F401
        handle = ctypes.CDLL(LIB)
        for name, (argtypes, restype) in SIGNATURES.items():
            fn = getattr(handle, name)  # AttributeError if include/simpb_hip.h and the .so disagree
            fn.argtypes = argtypes
            fn.restype = restype
        _lib = handle
    return _lib


def check(status, what):
    if status != 0:
        detail = ""
        if status == 2:
            detail = " - " + (lib().simpb_last_error() or b"").decode()
        raise RuntimeError(f"{what} failed: {ERRORS.get(status, status)}{detail}")
