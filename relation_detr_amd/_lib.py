"""ctypes binding of librelation_detr_amd.so (the C ABI declared in include/relation_detr_amd.h).

There is NO fallback: if the shared library is missing or a symbol is absent, importing the ops
raises.  Build it with ``python -c "import __graft_entry__ as g; g.build()"`` or
``make -C relation_detr_amd/csrc``.
"""
from __future__ import annotations

import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("RDETR_LIB_PATH") or os.path.join(_HERE, "librelation_detr_amd.so")    # override: A/B builds

_c_int, _c_float, _vp, _c_ll = ctypes.c_int, ctypes.c_float, ctypes.c_void_p, ctypes.c_longlong

# name -> argtypes (restype is int unless noted); mirrors include/relation_detr_amd.h one to one.
SIGNATURES = {
    "rdetr_abi_version": [],
    "rdetr_status_string": [_c_int],
    "rdetr_msda_fast_path": [_c_int] * 4,
    "rdetr_msda_levels_window_ok": [_vp, _vp, _c_int, _c_ll],
    "rdetr_msda_forward_f32": [_vp] * 5 + [_c_int] * 7 + [_vp, _vp],
    "rdetr_msda_forward_bf16": [_vp] * 5 + [_c_int] * 7 + [_vp, _vp],
    "rdetr_msda_forward_fused_f32": [_vp] * 6 + [_c_int] * 8 + [_vp, _vp],
    "rdetr_msda_forward_fused_bf16": [_vp] * 6 + [_c_int] * 8 + [_vp, _vp],
    "rdetr_msda_forward_fused_ex_f32": [_vp] * 4 + [_c_int, _vp, _c_int, _vp, _c_int, _vp] + [_c_int] * 7 + [_vp, _vp],
    "rdetr_msda_forward_fused_ex_bf16": [_vp] * 4 + [_c_int, _vp, _c_int, _vp, _c_int, _vp] + [_c_int] * 7 + [_vp, _vp],
    "rdetr_msda_forward_opt_bf16": [_vp, _c_int] + [_vp] * 4 + [_c_int] * 8 + [_vp, _vp],
    "rdetr_msda_forward_sweep_bf16": [_vp, _c_int] + [_vp] * 4 + [_c_int] * 7 + [_vp, _vp],
    "rdetr_msda_forward_resident_bf16": [_vp] * 5 + [_c_int] * 7 + [_vp, _vp],
    "rdetr_msda_forward_fused_resident_bf16": [_vp, _vp, _vp, _vp, _c_int, _vp, _c_int, _vp, _c_int] + [_c_int] * 7 + [_vp, _vp],
    "rdetr_msda_forward_fused_opt_bf16": [_vp, _c_int, _vp, _vp, _vp, _c_int, _vp, _c_int, _vp, _c_int, _vp] + [_c_int] * 8 + [_vp, _vp],
    "rdetr_msda_forward_fused_strided_bf16": [_vp, _c_ll, _vp, _vp, _vp, _c_int, _vp, _c_int, _vp, _c_int, _vp] + [_c_int] * 7 + [_vp, _vp],
    "rdetr_value_to_head_major_bf16": [_vp, _c_ll, _vp] + [_c_int] * 4 + [_vp, _vp],
    "rdetr_msda_backward_f32": [_vp] * 6 + [_c_int] * 7 + [_vp] * 4,
    "rdetr_msda_backward_det_workspace_bytes": [_c_int] * 7,
    "rdetr_msda_backward_det_f32": [_vp] * 6 + [_c_int] * 7 + [_vp, _c_ll] + [_vp] * 4,
    "rdetr_relation_bias_f32": [_vp] * 4 + [_c_int] * 5 + [_c_float] * 3 + [_vp, _vp],
    "rdetr_relation_bias_ws_f32": [_vp] * 4 + [_c_int] * 5 + [_c_float] * 3 + [_vp, _vp, _vp],
    "rdetr_relation_bias_backward_workspace_bytes": [_c_int] * 3,
    "rdetr_relation_bias_backward_f32": [_vp] * 4 + [_c_int] * 5 + [_c_float] * 3 + [_vp] * 4,
    "rdetr_bias_softmax_f32": [_vp] * 3 + [_c_int] * 3 + [_vp],
    "rdetr_relation_attention_bf16": [_vp] * 3 + [_c_int] * 3 + [_vp, _vp] + [_c_int] * 5 + [_c_float, _vp, _c_int, _vp],
    "rdetr_relation_attention_boxes_bf16": [_vp] * 3 + [_c_int] * 3 + [_vp] * 5 + [_c_int] * 6 + [_c_float] * 4 + [_vp, _c_int, _vp],
    "rdetr_box_refine_f32": [_vp, _c_int, _vp, _c_ll, _c_float, _vp, _vp],
    "rdetr_sine_pos_embed": [_vp, _c_ll, _c_int, _c_int, _c_float, _c_float, _vp, _c_int, _vp],
    "rdetr_zero_masked_rows": [_vp, _vp, _c_ll, _c_int, _c_ll, _vp],
    "rdetr_row_max": [_vp, _c_int, _c_ll, _c_int, _c_ll, _vp, _vp],
    "rdetr_topk_workspace_bytes": [_c_int, _c_int, _c_int],
    "rdetr_topk": [_vp, _c_int, _c_int, _c_int, _c_int, _vp, _vp, _vp, _vp],
    "rdetr_box_head_k256_bf16": [_vp, _c_ll, _vp, _c_ll, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _c_int, _c_float, _c_ll, _vp, _vp, _vp],
    "rdetr_query_pos_k256_bf16": [_vp, _c_ll, _vp, _c_ll] + [_vp] * 9 + [_c_ll, _vp, _vp, _vp],
    "rdetr_encoder_proj_k256_bf16": [_vp, _c_ll, _vp, _c_ll] + [_vp] * 5 + [_c_int, _c_int, _c_int, _vp, _vp, _vp],
    "rdetr_detections_from_topk": [_vp, _vp, _vp, _vp, _c_int, _c_int, _c_int, _c_int, _vp, _vp],
    "rdetr_scaled_pos": [_vp, _vp, _vp, _c_ll, _c_int, _vp, _vp, _vp],
    "rdetr_decoder_reference": [_vp, _vp, _c_int, _c_int, _c_int, _c_int, _c_float, _c_float, _vp, _vp, _c_int, _vp],
    "rdetr_pyramid_points": [_vp, _vp, _c_int, _c_int, _vp, _c_int, _vp, _vp, _vp, _vp, _vp],
    "rdetr_linear_k256_bf16": [_vp, _c_ll, _vp, _vp, _c_ll, _c_int, _c_int, _vp, _c_ll, _vp],
    "rdetr_linear_k256_hm_bf16": [_vp, _c_ll, _vp, _vp, _vp, _c_int, _c_int, _vp, _vp],
    "rdetr_linear_pack_k256_bf16": [_vp, _vp, _vp],
    "rdetr_linear_ln_k256_bf16": [_vp, _c_ll, _vp, _vp, _vp, _c_ll, _vp, _vp, _c_float, _c_ll, _vp, _c_ll, _vp],
    "rdetr_ffn_k256_pack_bf16": [_vp, _vp, _c_int, _vp, _vp],
    "rdetr_ffn_k256_bf16": [_vp, _c_ll, _vp, _vp, _vp, _c_ll, _c_int, _vp, _c_ll, _vp],
    "rdetr_ffn_ln_k256_bf16": [_vp, _c_ll, _vp, _vp, _vp, _vp, _vp, _c_float, _vp, _c_ll, _c_ll, _c_int, _vp, _c_ll, _vp, _c_ll, _vp],
    "rdetr_nchw_to_tokens": [_vp, _vp, _c_int, _c_int, _c_int, _c_int, _c_ll, _c_ll, _vp, _vp],
    "rdetr_add_layernorm_f32": [_vp] * 4 + [_c_ll, _c_int, _c_float, _vp, _vp],
    "rdetr_add_layernorm_bf16": [_vp] * 4 + [_c_ll, _c_int, _c_float, _vp, _vp],
    "rdetr_add_layernorm_strided_f32": [_vp] * 4 + [_c_ll, _c_int, _c_ll, _c_ll, _c_ll, _c_float, _vp, _vp],
    "rdetr_add_layernorm_strided_bf16": [_vp] * 4 + [_c_ll, _c_int, _c_ll, _c_ll, _c_ll, _c_float, _vp, _vp],
    "rdetr_add_layernorm_pos_f32": [_vp] * 5 + [_c_ll, _c_int] + [_c_ll] * 5 + [_c_float, _vp, _vp, _vp],
    "rdetr_add_layernorm_pos_bf16": [_vp] * 5 + [_c_ll, _c_int] + [_c_ll] * 5 + [_c_float, _vp, _vp, _vp],
}

ERR_INVALID_ARG, ERR_UNSUPPORTED, ERR_LAUNCH = -1, -2, -3            # rdetr_status (include/relation_detr_amd.h)

_lib = None


class RdetrError(RuntimeError):
    pass


def load() -> ctypes.CDLL:
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RdetrError(
            f"{LIB_PATH} not found: the HIP kernels are not built. Run `make -C relation_detr_amd/csrc` "
            "(needs hipcc, --offload-arch=gfx950). There is no CPU or PyTorch fallback for this path.")
    lib = ctypes.CDLL(LIB_PATH)
    for name, argtypes in SIGNATURES.items():
        fn = getattr(lib, name)          # AttributeError if the .so is stale
        fn.argtypes = argtypes
        fn.restype = ctypes.c_char_p if name == "rdetr_status_string" else (_c_ll if name.endswith("_workspace_bytes") else _c_int)
    if lib.rdetr_abi_version() != 3:
        raise RdetrError(f"ABI version mismatch: library {lib.rdetr_abi_version()}, binding 3")
    _lib = lib
    return lib


def check(status: int, what: str) -> None:
    if status != 0:
        msg = load().rdetr_status_string(status).decode()
        raise RdetrError(f"{what} failed: {msg} (status {status})")
