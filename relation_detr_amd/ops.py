"""Operator layer: torch tensors in, HIP kernels (through the C ABI) out.

Mirrors the reference's pybind module ``_C`` (models/bricks/ops/cuda/ms_deform_attn_cuda.cu:148-151):
``ms_deform_attn_forward`` / ``ms_deform_attn_backward`` keep the reference's argument order and
meaning, allocate and return new tensors, require contiguous device tensors and run on the current
stream.  ``relation_bias`` and ``bias_softmax_`` are the two relation-path operators.

No CPU path: a tensor that is not on a ROCm device raises, and so does a missing library.
"""
from __future__ import annotations

import collections
import weakref
from typing import Optional, Tuple

import torch

from . import _lib


_KEEP: "collections.deque" = collections.deque(maxlen=64)


def _cptr(t: torch.Tensor) -> int:
    """Device pointer of ``t`` made contiguous.  A copy made here must outlive the Python expression that asked for it -- the
    launch that reads it is enqueued only after ALL arguments are evaluated, and a later argument may allocate -- so the last
    few copies are kept referenced (small vectors: biases, LayerNorm parameters; contiguous tensors are passed through)."""
    if t.is_contiguous():
        return t.data_ptr()
    c = t.contiguous()
    _KEEP.append(c)
    return c.data_ptr()


def _stream_ptr(t: torch.Tensor) -> int:
    return torch.cuda.current_stream(t.device).cuda_stream


def _require_device(*tensors: torch.Tensor) -> None:
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise _lib.RdetrError(
                "relation_detr_amd operators need tensors on a ROCm device (got a CPU tensor); "
                "there is no CPU fallback on the product path")


def _require_contiguous(**named: torch.Tensor) -> None:
    for name, t in named.items():
        if t is not None and not t.is_contiguous():
            raise _lib.RdetrError(f"{name} tensor has to be contiguous")      # AT_ASSERTM, ms_deform_attn_cuda.cu:20-24


# Host copies of (spatial_shapes, level_start_index), cached per tensor OBJECT (weak references, so
# a recycled device address can never alias a stale entry): one D2H sync per new pyramid instead of
# one per call (the reference syncs on every call, ms_deform_attn.py:313).
_shape_cache: dict = {}


def host_levels(spatial_shapes: torch.Tensor, level_start_index: torch.Tensor) -> Tuple[tuple, tuple]:
    key = (id(spatial_shapes), id(level_start_index))
    hit = _shape_cache.get(key)
    if hit is not None:
        ref_s, ref_l, ver, levels = hit
        if ref_s() is spatial_shapes and ref_l() is level_start_index and ver == (spatial_shapes._version,
                                                                                 level_start_index._version):
            return levels
    if len(_shape_cache) > 64:
        _shape_cache.clear()
    levels = (tuple(map(tuple, spatial_shapes.tolist())), tuple(level_start_index.tolist()))
    _shape_cache[key] = (weakref.ref(spatial_shapes), weakref.ref(level_start_index),
                         (spatial_shapes._version, level_start_index._version), levels)
    return levels


def check_levels(spatial_shapes: torch.Tensor, level_start_index: torch.Tensor, num_value: int) -> None:
    """Host-side validation that the pyramid description matches the value tensor, so that the
    kernel's indexing assumptions hold before anything is launched."""
    if spatial_shapes.dim() != 2 or spatial_shapes.shape[1] != 2 or spatial_shapes.dtype != torch.int64:
        raise _lib.RdetrError("spatial_shapes must be an int64 tensor of shape [L, 2]")
    if level_start_index.dtype != torch.int64 or level_start_index.numel() != spatial_shapes.shape[0]:
        raise _lib.RdetrError("level_start_index must be an int64 tensor of shape [L]")
    shapes, starts = host_levels(spatial_shapes, level_start_index)
    for (h, w), s in zip(shapes, starts):
        if h <= 0 or w <= 0 or s < 0 or s + h * w > num_value:
            raise _lib.RdetrError(
                f"level (h={h}, w={w}, start={s}) does not fit a value tensor with {num_value} positions")


VALUE_BSHD, VALUE_BHSD = 0, 1                     # include/relation_detr_amd.h: RDETR_VALUE_*
MSDA_AUTO, MSDA_DIRECT, MSDA_WINDOW, MSDA_AUTO_PACKED = 0, 1, 2, 3     # RDETR_MSDA_*


_window_ok_cache: dict = {}


def levels_window_ok(spatial_shapes: torch.Tensor, level_start_index: torch.Tensor, num_value: int) -> bool:
    """Precondition of the LDS-window kernel (include/relation_detr_amd.h, RDETR_MSDA_WINDOW): the levels tile [0, S) exactly
    -- cumulative starts, sum(h*w) == S -- and none outgrows level 0.  From the cached host copy of the table, through the
    library's own host helper (one definition of the rule)."""
    import ctypes
    shapes, starts = host_levels(spatial_shapes, level_start_index)
    key = (shapes, starts, int(num_value))
    hit = _window_ok_cache.get(key)                     # a few distinct pyramids per process: decided once each
    if hit is None:
        n = len(shapes)
        hs = (ctypes.c_int64 * (2 * n))(*[v for hw in shapes for v in hw])
        st = (ctypes.c_int64 * n)(*starts)
        hit = bool(_lib.load().rdetr_msda_levels_window_ok(hs, st, n, num_value))
        if len(_window_ok_cache) > 256:
            _window_ok_cache.clear()
        _window_ok_cache[key] = hit
    return hit


_host_array_cache: dict = {}


def _host_level_arrays(spatial_shapes: torch.Tensor, level_start_index: torch.Tensor):
    """ctypes int64 arrays of the cached host copy of the level table (the tile kernel's entry points take HOST pointers)."""
    import ctypes
    key = host_levels(spatial_shapes, level_start_index)
    hit = _host_array_cache.get(key)
    if hit is None:
        shapes, starts = key
        hit = ((ctypes.c_int64 * (2 * len(shapes)))(*[v for hw in shapes for v in hw]), (ctypes.c_int64 * len(starts))(*starts))
        if len(_host_array_cache) > 256:
            _host_array_cache.clear()
        _host_array_cache[key] = hit
    return hit


def _msda_algo(algo: str, spatial_shapes, level_start_index, num_value: int) -> int:
    """'auto' may take the window kernel only for a level table that meets its precondition; an explicit 'window' on one
    that does not is refused (it would leave output rows unwritten)."""
    if algo == "direct":
        return MSDA_DIRECT
    ok = levels_window_ok(spatial_shapes, level_start_index, num_value)
    if algo == "window":
        if not ok:
            raise _lib.RdetrError("algo='window' needs levels that tile [0, S) exactly (cumulative level_start_index, "
                                  "sum(h*w) == S, no level larger than level 0)")
        return MSDA_WINDOW
    if algo != "auto":
        raise ValueError("algo must be 'auto', 'direct' or 'window'")
    return MSDA_AUTO_PACKED if ok else MSDA_DIRECT


def _resident_pays(B: int, Nq: int, L: int, level_shapes=None) -> bool:
    """'auto' on a head-major bf16 value: the resident-levels kernel (csrc/msda_res.hip: one persistent workgroup per CU) where it
    was measured faster than the query-run kernel -- four levels, the last TWO of them (half of all samples) fitting the CU's LDS
    beside the staging area, and enough runs of 16 queries to feed 256 workgroups."""
    if L != 4 or B * Nq < 16384:
        return False
    if level_shapes is None:
        return True
    coarse = sum(int(h) * int(w) for h, w in level_shapes[2:]) * 64          # bytes of levels 2 and 3 of one (image, head) plane
    return coarse <= 160 * 1024 - 128 - 12 * 4096 - 128                       # msda_res.hip: zero row + 12 waves' staging + alignment


def _value_dims(value: torch.Tensor, layout: str):
    """(B, S, H, D) of a value tensor in layout "bshd" ([B,S,H,D], the reference operator's) or "bhsd" (head-major)."""
    if value.dim() != 4 or layout not in ("bshd", "bhsd"):
        raise _lib.RdetrError("expected a 4-d value tensor in layout 'bshd' or 'bhsd'")
    if layout == "bshd":
        return tuple(value.shape)
    B, H, S, D = value.shape
    return B, S, H, D


def ms_deform_attn_forward(value: torch.Tensor, spatial_shapes: torch.Tensor, level_start_index: torch.Tensor,
                           sampling_loc: torch.Tensor, attn_weight: torch.Tensor, im2col_step: int = 64,
                           value_layout: str = "bshd", algo: str = "auto") -> torch.Tensor:
    """value [B,S,H,D] (fp32 or bf16), sampling_loc [B,Nq,H,L,P,2] fp32, attn_weight [B,Nq,H,L,P] fp32
    -> [B,Nq,H*D] in value's dtype.  ``im2col_step`` is accepted and ignored (no batch restriction).
    Not in the reference's signature (optional, bf16 only): ``value_layout="bhsd"`` for a head-major value [B,H,S,D]
    (`value_to_head_major`), ``algo`` "auto" | "direct" | "window" | "sweep" | "resident" to name the kernel (tests, A/B timing;
    "sweep" = csrc/msda_sweep.hip, opt-in; "resident" = csrc/msda_res.hip, what "auto" takes for a head-major value and a large
    query count; both take the cached HOST copy of the level table)."""
    _require_device(value, spatial_shapes, level_start_index, sampling_loc, attn_weight)
    _require_contiguous(value=value, spatial_shapes=spatial_shapes, level_start_index=level_start_index,
                        sampling_loc=sampling_loc, attn_weight=attn_weight)
    if value.dim() != 4 or sampling_loc.dim() != 6 or attn_weight.dim() != 5:
        raise _lib.RdetrError("expected value [B,S,H,D], sampling_loc [B,Nq,H,L,P,2], attn_weight [B,Nq,H,L,P]")
    B, S, H, D = _value_dims(value, value_layout)
    _, Nq, H2, L, P, two = sampling_loc.shape
    if (H2, two) != (H, 2) or tuple(attn_weight.shape) != (B, Nq, H, L, P) or sampling_loc.shape[0] != B:
        raise _lib.RdetrError("sampling_loc / attn_weight shapes do not match value")
    if spatial_shapes.shape[0] != L:
        raise _lib.RdetrError("spatial_shapes has a different number of levels than sampling_loc")
    if sampling_loc.dtype != torch.float32 or attn_weight.dtype != torch.float32:
        raise _lib.RdetrError("sampling_loc and attn_weight must be float32")
    if algo not in ("auto", "direct", "window", "sweep", "resident"):
        raise ValueError("algo must be 'auto', 'direct', 'window', 'sweep' or 'resident'")
    check_levels(spatial_shapes, level_start_index, S)
    lib = _lib.load()
    out = torch.empty(B, Nq, H * D, dtype=value.dtype, device=value.device)
    if value.dtype == torch.bfloat16 and algo == "sweep":
        hs, st_h = _host_level_arrays(spatial_shapes, level_start_index)
        st = lib.rdetr_msda_forward_sweep_bf16(value.data_ptr(), VALUE_BHSD if value_layout == "bhsd" else VALUE_BSHD, hs, st_h,
                                               sampling_loc.data_ptr(), attn_weight.data_ptr(), B, S, H, D, L, Nq, P,
                                               out.data_ptr(), _stream_ptr(value))
        _lib.check(st, "rdetr_msda_forward_sweep_bf16")
        return out
    if value.dtype == torch.bfloat16 and value_layout == "bhsd" and (algo == "resident" or (algo == "auto" and _resident_pays(B, Nq, L, host_levels(spatial_shapes, level_start_index)[0]))):
        # persistent workgroups with the coarse levels resident in LDS (csrc/msda_res.hip): large query counts on the head-major
        # layout; anything it does not cover comes back as RDETR_ERR_UNSUPPORTED and runs on the query-run kernel below
        hs, st_h = _host_level_arrays(spatial_shapes, level_start_index)
        st = lib.rdetr_msda_forward_resident_bf16(value.data_ptr(), hs, st_h, sampling_loc.data_ptr(), attn_weight.data_ptr(),
                                                  B, S, H, D, L, Nq, P, out.data_ptr(), _stream_ptr(value))
        if st != _lib.ERR_UNSUPPORTED or algo == "resident":
            _lib.check(st, "rdetr_msda_forward_resident_bf16")
            return out
    if algo == "resident":
        raise _lib.RdetrError("algo='resident' needs a bfloat16 head-major value (value_layout='bhsd')")
    if value.dtype == torch.bfloat16:
        st = lib.rdetr_msda_forward_opt_bf16(value.data_ptr(), VALUE_BHSD if value_layout == "bhsd" else VALUE_BSHD,
                                             spatial_shapes.data_ptr(), level_start_index.data_ptr(), sampling_loc.data_ptr(),
                                             attn_weight.data_ptr(), B, S, H, D, L, Nq, P,
                                             _msda_algo(algo, spatial_shapes, level_start_index, S), out.data_ptr(),
                                             _stream_ptr(value))
        _lib.check(st, "rdetr_msda_forward_opt_bf16")
        return out
    if value_layout != "bshd" or algo != "auto":
        raise _lib.RdetrError("value_layout / algo options exist for bfloat16 value only")
    if value.dtype != torch.float32:
        raise _lib.RdetrError(f"value dtype {value.dtype} not supported (float32 or bfloat16)")
    st = lib.rdetr_msda_forward_f32(value.data_ptr(), spatial_shapes.data_ptr(), level_start_index.data_ptr(),
                                    sampling_loc.data_ptr(), attn_weight.data_ptr(), B, S, H, D, L, Nq, P, out.data_ptr(),
                                    _stream_ptr(value))
    _lib.check(st, "rdetr_msda_forward_f32")
    return out


def value_to_head_major(value: torch.Tensor, key_padding_mask: Optional[torch.Tensor] = None) -> torch.Tensor:
    """Projected value [B,S,256] bf16 (rows may be a column slice of a wider buffer) -> head-major [B,8,S,32], the rows of
    padded positions zeroed on the way (ms_deform_attn.py:316-319): the layout the window kernel fills its LDS windows
    from at the contiguous-row rate (pass the result with ``value_layout="bhsd"``)."""
    _require_device(value, key_padding_mask)
    if value.dim() != 3 or value.shape[-1] != 256 or value.dtype != torch.bfloat16:
        raise _lib.RdetrError("value_to_head_major: expected a bfloat16 [B, S, 256] tensor")
    B, S, _ = value.shape
    if value.stride(2) != 1 or (B > 1 and value.stride(0) != S * value.stride(1)) or value.stride(1) % 8 or value.data_ptr() % 16:
        value = value.contiguous()
    mask_ptr = None
    if key_padding_mask is not None:
        if tuple(key_padding_mask.shape) != (B, S):
            raise _lib.RdetrError("key_padding_mask must be [B, S]")
        mask_u8 = key_padding_mask.contiguous().view(torch.uint8) if key_padding_mask.dtype == torch.bool \
            else key_padding_mask.to(torch.uint8).contiguous()
        mask_ptr = mask_u8.data_ptr()
    out = torch.empty(B, 8, S, 32, dtype=torch.bfloat16, device=value.device)
    st = _lib.load().rdetr_value_to_head_major_bf16(value.data_ptr(), value.stride(1), mask_ptr, B, S, 8, 32, out.data_ptr(),
                                                    _stream_ptr(value))
    _lib.check(st, "rdetr_value_to_head_major_bf16")
    return out


def _producer_row_stride(t: torch.Tensor):
    """0 for a contiguous tensor; the row stride (elements) if `t` [B, Nq, ...] is a column slice of a wider row-major
    matrix (every row contiguous, rows evenly strided, even stride); None if neither."""
    if t.is_contiguous():
        return 0
    inner = 1
    for d in range(t.dim() - 1, 1, -1):                  # dims after (B, Nq) must be contiguous among themselves
        if t.shape[d] != 1 and t.stride(d) != inner:
            return None
        inner *= t.shape[d]
    ld = t.stride(1)
    if ld < inner or ld % 2 or (t.shape[0] > 1 and t.stride(0) != t.shape[1] * ld):
        return None
    return ld


def ms_deform_attn_forward_fused(value: torch.Tensor, spatial_shapes: torch.Tensor, level_start_index: torch.Tensor,
                                 sampling_offsets: torch.Tensor, attn_logits: torch.Tensor,
                                 reference_points: torch.Tensor, key_padding_mask: Optional[torch.Tensor] = None,
                                 value_layout: str = "bshd", algo: str = "auto") -> torch.Tensor:
    """MSDA with the location / weight producer fused into the gather kernel (inference path).
    value [B,S,H,D] fp32|bf16 (or head-major [B,H,S,D] bf16 with ``value_layout="bhsd"``); sampling_offsets
    [B,Nq,H,L,P,2] and attn_logits [B,Nq,H,L*P] RAW projection outputs in value's dtype; reference_points [B,Nq,L,2|4] fp32
    -> [B,Nq,H*D] in value's dtype.  ``key_padding_mask`` [B,S]: `value` is then the UNFILLED projection and the kernel treats
    the rows of padded positions as zero.
    Same result as softmax + sampling-location arithmetic + ms_deform_attn_forward (ms_deform_attn.py:322-370)."""
    _require_device(value, spatial_shapes, level_start_index, sampling_offsets, attn_logits, reference_points, key_padding_mask)
    # the two projection outputs may be column slices of one wider GEMM output (rows evenly strided, each row contiguous)
    ld_off = _producer_row_stride(sampling_offsets)
    ld_lg = _producer_row_stride(attn_logits)
    if ld_off is None:
        sampling_offsets, ld_off = sampling_offsets.contiguous(), 0
    if ld_lg is None:
        attn_logits, ld_lg = attn_logits.contiguous(), 0
    # a [B,S,H,D] bf16 value may be row-strided: the 256-column slice of a wider projection output (pixel stride in elements)
    value_ld = 0
    if (value.dim() == 4 and not value.is_contiguous() and value_layout == "bshd" and value.dtype == torch.bfloat16
            and value.stride(3) == 1 and value.stride(2) == value.shape[3] and value.stride(1) >= value.shape[2] * value.shape[3]
            and value.stride(1) % 8 == 0 and (value.shape[0] == 1 or value.stride(0) == value.shape[1] * value.stride(1))
            and value.data_ptr() % 16 == 0 and algo in ("auto", "direct")):
        value_ld = value.stride(1)
    else:
        _require_contiguous(value=value)
    _require_contiguous(spatial_shapes=spatial_shapes, level_start_index=level_start_index, reference_points=reference_points)
    if value.dim() != 4 or sampling_offsets.dim() != 6 or reference_points.dim() != 4:
        raise _lib.RdetrError("expected value [B,S,H,D], sampling_offsets [B,Nq,H,L,P,2], reference_points [B,Nq,L,2|4]")
    B, S, H, D = _value_dims(value, value_layout)
    _, Nq, H2, L, P, two = sampling_offsets.shape
    ref_dim = reference_points.shape[-1]
    if (H2, two) != (H, 2) or sampling_offsets.shape[0] != B or attn_logits.numel() != B * Nq * H * L * P:
        raise _lib.RdetrError("sampling_offsets / attn_logits shapes do not match value")
    if tuple(reference_points.shape[:3]) != (B, Nq, L) or ref_dim not in (2, 4):
        raise ValueError("Last dim of reference_points must be 2 or 4, but get {} instead.".format(ref_dim))
    if sampling_offsets.dtype != value.dtype or attn_logits.dtype != value.dtype or reference_points.dtype != torch.float32:
        raise _lib.RdetrError("sampling_offsets / attn_logits must have value's dtype, reference_points float32")
    if spatial_shapes.shape[0] != L:
        raise _lib.RdetrError("spatial_shapes has a different number of levels than sampling_offsets")
    if algo not in ("auto", "direct", "window", "resident"):
        raise ValueError("algo must be 'auto', 'direct', 'window' or 'resident'")
    if value.dtype not in (torch.float32, torch.bfloat16):
        raise _lib.RdetrError(f"value dtype {value.dtype} not supported (float32 or bfloat16)")
    if value.dtype == torch.float32 and (value_layout != "bshd" or algo != "auto"):
        raise _lib.RdetrError("value_layout / algo options exist for bfloat16 value only")
    check_levels(spatial_shapes, level_start_index, S)
    lib = _lib.load()
    mask_ptr = None
    if key_padding_mask is not None:
        if tuple(key_padding_mask.shape) != (B, S):
            raise _lib.RdetrError("key_padding_mask must be [B, S]")
        mask_u8 = key_padding_mask.contiguous().view(torch.uint8) if key_padding_mask.dtype == torch.bool \
            else key_padding_mask.to(torch.uint8).contiguous()
        mask_ptr = mask_u8.data_ptr()
    out = torch.empty(B, Nq, H * D, dtype=value.dtype, device=value.device)
    if (value.dtype == torch.bfloat16 and value_layout == "bhsd" and mask_ptr is None and not value_ld
            and (algo == "resident" or (algo == "auto" and _resident_pays(B, Nq, L, host_levels(spatial_shapes, level_start_index)[0])))):
        hs, st_h = _host_level_arrays(spatial_shapes, level_start_index)
        st = lib.rdetr_msda_forward_fused_resident_bf16(
            value.data_ptr(), hs, st_h, sampling_offsets.data_ptr(), ld_off, attn_logits.data_ptr(), ld_lg,
            reference_points.data_ptr(), ref_dim, B, S, H, D, L, Nq, P, out.data_ptr(), _stream_ptr(value))
        if st != _lib.ERR_UNSUPPORTED or algo == "resident":
            _lib.check(st, "rdetr_msda_forward_fused_resident_bf16")
            return out
    if algo == "resident":
        raise _lib.RdetrError("algo='resident' needs a bfloat16 head-major value (value_layout='bhsd') and no key_padding_mask")
    if value_ld:
        st = lib.rdetr_msda_forward_fused_strided_bf16(
            value.data_ptr(), value_ld, spatial_shapes.data_ptr(), level_start_index.data_ptr(), sampling_offsets.data_ptr(), ld_off,
            attn_logits.data_ptr(), ld_lg, reference_points.data_ptr(), ref_dim, mask_ptr, B, S, H, D, L, Nq, P, out.data_ptr(),
            _stream_ptr(value))
        _lib.check(st, "rdetr_msda_forward_fused_strided_bf16")
        return out
    if value.dtype == torch.bfloat16:
        st = lib.rdetr_msda_forward_fused_opt_bf16(
            value.data_ptr(), VALUE_BHSD if value_layout == "bhsd" else VALUE_BSHD, spatial_shapes.data_ptr(),
            level_start_index.data_ptr(), sampling_offsets.data_ptr(), ld_off, attn_logits.data_ptr(), ld_lg,
            reference_points.data_ptr(), ref_dim, mask_ptr, B, S, H, D, L, Nq, P,
            _msda_algo(algo, spatial_shapes, level_start_index, S), out.data_ptr(), _stream_ptr(value))
        _lib.check(st, "rdetr_msda_forward_fused_opt_bf16")
        return out
    if ld_off or ld_lg or mask_ptr is not None:
        st = lib.rdetr_msda_forward_fused_ex_f32(value.data_ptr(), spatial_shapes.data_ptr(), level_start_index.data_ptr(),
                                                 sampling_offsets.data_ptr(), ld_off, attn_logits.data_ptr(), ld_lg,
                                                 reference_points.data_ptr(), ref_dim, mask_ptr, B, S, H, D, L, Nq, P,
                                                 out.data_ptr(), _stream_ptr(value))
        _lib.check(st, "rdetr_msda_forward_fused_ex_f32")
        return out
    st = lib.rdetr_msda_forward_fused_f32(value.data_ptr(), spatial_shapes.data_ptr(), level_start_index.data_ptr(),
                                          sampling_offsets.data_ptr(), attn_logits.data_ptr(), reference_points.data_ptr(),
                                          ref_dim, B, S, H, D, L, Nq, P, out.data_ptr(), _stream_ptr(value))
    _lib.check(st, "rdetr_msda_forward_fused_f32")
    return out


def msda_fast_path(H: int, D: int, L: int, P: int) -> bool:
    return bool(_lib.load().rdetr_msda_fast_path(H, D, L, P))


def ms_deform_attn_backward(value: torch.Tensor, spatial_shapes: torch.Tensor, level_start_index: torch.Tensor,
                            sampling_loc: torch.Tensor, attn_weight: torch.Tensor, grad_output: torch.Tensor,
                            im2col_step: int = 64, deterministic: Optional[bool] = None):
    """-> [grad_value, grad_sampling_loc, grad_attn_weight] (fp32), as ms_deform_attn_cuda.cu:75-145.
    ``deterministic`` (not in the reference's signature; default = ``torch.are_deterministic_algorithms_enabled()``): grad_value
    through sorted per-row sums instead of float atomics (csrc/msda_bwd.hip; H = 8, D = 32, P = 4) -- the same bits on every run."""
    _require_device(value, spatial_shapes, level_start_index, sampling_loc, attn_weight, grad_output)
    grad_output = grad_output.contiguous()
    _require_contiguous(value=value, spatial_shapes=spatial_shapes, level_start_index=level_start_index,
                        sampling_loc=sampling_loc, attn_weight=attn_weight)
    if value.dtype != torch.float32 or grad_output.dtype != torch.float32:
        raise _lib.RdetrError("ms_deform_attn_backward is float32 only")
    B, S, H, D = value.shape
    _, Nq, _, L, P, _ = sampling_loc.shape
    if tuple(grad_output.shape) != (B, Nq, H * D):
        raise _lib.RdetrError("grad_output must be [B, Nq, H*D]")
    check_levels(spatial_shapes, level_start_index, S)
    if deterministic is None:
        deterministic = torch.are_deterministic_algorithms_enabled()
    if deterministic:
        lib = _lib.load()
        nbytes = int(lib.rdetr_msda_backward_det_workspace_bytes(B, S, H, D, L, Nq, P))
        if nbytes < 0 or (nbytes == 0 and B * S * Nq > 0):
            raise _lib.RdetrError("deterministic ms_deform_attn_backward: H = 8, D = 32, P = 4, L <= 8 and fewer than 2^31 sample corners only")
        ws = torch.empty(max(nbytes, 16), dtype=torch.uint8, device=value.device)
        grad_value = torch.empty_like(value)                    # every row is written
        grad_loc = torch.empty_like(sampling_loc)
        grad_attn = torch.empty_like(attn_weight)
        st = lib.rdetr_msda_backward_det_f32(
            value.data_ptr(), spatial_shapes.data_ptr(), level_start_index.data_ptr(), sampling_loc.data_ptr(), attn_weight.data_ptr(),
            grad_output.data_ptr(), B, S, H, D, L, Nq, P, ws.data_ptr(), nbytes, grad_value.data_ptr(), grad_loc.data_ptr(),
            grad_attn.data_ptr(), _stream_ptr(value))
        _lib.check(st, "rdetr_msda_backward_det_f32")
        return [grad_value, grad_loc, grad_attn]
    grad_value = torch.zeros_like(value)                       # accumulated with atomics
    grad_loc = torch.empty_like(sampling_loc)
    grad_attn = torch.empty_like(attn_weight)
    st = _lib.load().rdetr_msda_backward_f32(
        value.data_ptr(), spatial_shapes.data_ptr(), level_start_index.data_ptr(), sampling_loc.data_ptr(),
        attn_weight.data_ptr(), grad_output.data_ptr(), B, S, H, D, L, Nq, P, grad_value.data_ptr(),
        grad_loc.data_ptr(), grad_attn.data_ptr(), _stream_ptr(value))
    _lib.check(st, "rdetr_msda_backward_f32")
    return [grad_value, grad_loc, grad_attn]


class MultiScaleDeformableAttnFunction(torch.autograd.Function):
    """Autograd wrapper with the reference's signature (models/bricks/ms_deform_attn.py:35-84)."""

    @staticmethod
    def forward(ctx, value, value_spatial_shapes, value_level_start_index, sampling_locations, attention_weights,
                im2col_step=64):
        ctx.im2col_step = im2col_step
        out = ms_deform_attn_forward(value, value_spatial_shapes, value_level_start_index, sampling_locations,
                                     attention_weights, im2col_step)
        ctx.save_for_backward(value, value_spatial_shapes, value_level_start_index, sampling_locations,
                              attention_weights)
        return out

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, grad_output):
        value, shapes, starts, loc, attn = ctx.saved_tensors
        gv, gl, ga = ms_deform_attn_backward(value.float(), shapes, starts, loc, attn, grad_output.float(),
                                             ctx.im2col_step)
        return gv.to(value.dtype), None, None, gl, ga, None


def relation_bias(src_boxes: torch.Tensor, tgt_boxes: torch.Tensor, proj_weight: torch.Tensor,
                  proj_bias: Optional[torch.Tensor], num_pos_feats: int = 16, temperature: float = 10000.0,
                  scale: float = 100.0, eps: float = 1e-5) -> torch.Tensor:
    """boxes [B,N,4] cxcywh, proj_weight [Hh, 4F(,1,1)], proj_bias [Hh] -> ReLU(conv1x1(sine(rel))) [B,Hh,N1,N2]."""
    _require_device(src_boxes, tgt_boxes, proj_weight, proj_bias)
    if src_boxes.dim() != 3 or tgt_boxes.dim() != 3 or src_boxes.shape[0] != tgt_boxes.shape[0]:
        raise _lib.RdetrError("boxes must be [B, N, 4] with equal batch size")
    src = src_boxes.detach().float().contiguous()
    tgt = tgt_boxes.detach().float().contiguous()
    Hh = proj_weight.shape[0]
    w = proj_weight.detach().float().reshape(Hh, -1).contiguous()
    if w.shape[1] != 4 * num_pos_feats:
        raise _lib.RdetrError(f"proj_weight has {w.shape[1]} input channels, expected {4 * num_pos_feats}")
    b = None if proj_bias is None else proj_bias.detach().float().contiguous()
    B, N1, _ = src.shape
    N2 = tgt.shape[1]
    out = torch.empty(B, Hh, N1, N2, dtype=torch.float32, device=src.device)
    # workspace for the per-box sine tables of the size-ratio coordinates (used for F = 16, Hh = 8; ignored otherwise)
    ws = torch.empty((B * N1 + B * N2) * 2 * num_pos_feats, dtype=torch.float32, device=src.device)
    st = _lib.load().rdetr_relation_bias_ws_f32(src.data_ptr(), tgt.data_ptr(), w.data_ptr(),
                                                None if b is None else b.data_ptr(), B, N1, N2, Hh, num_pos_feats,
                                                scale, temperature, eps, ws.data_ptr(), out.data_ptr(), _stream_ptr(src))
    _lib.check(st, "rdetr_relation_bias_ws_f32")
    return out


def relation_bias_backward_supported(num_heads: int, num_pos_feats: int) -> bool:
    return num_heads == 8 and num_pos_feats == 16


def relation_bias_backward(src_boxes: torch.Tensor, tgt_boxes: torch.Tensor, grad_out: torch.Tensor, active: torch.Tensor,
                           num_pos_feats: int = 16, temperature: float = 10000.0, scale: float = 100.0, eps: float = 1e-5):
    """Gradients of ``relation_bias`` with respect to the projection: (grad_weight [Hh, 4F], grad_bias [Hh]) from the upstream
    gradient ``grad_out`` [B,Hh,N1,N2] and the ReLU mask ``active`` (bool, forward output > 0) -- the sine features are
    regenerated from the boxes inside the kernel (csrc/relation_bwd.hip), the reduction is deterministic.  Hh = 8, F = 16."""
    _require_device(src_boxes, tgt_boxes, grad_out, active)
    B, Hh, N1, N2 = grad_out.shape
    if not relation_bias_backward_supported(Hh, num_pos_feats):
        raise _lib.RdetrError("relation_bias_backward: 8 heads and 16 sine features per coordinate only")
    if tuple(src_boxes.shape) != (B, N1, 4) or tuple(tgt_boxes.shape) != (B, N2, 4) or active.shape != grad_out.shape:
        raise _lib.RdetrError("relation_bias_backward: src [B,N1,4], tgt [B,N2,4], grad_out / active [B,Hh,N1,N2]")
    src, tgt = src_boxes.detach().float().contiguous(), tgt_boxes.detach().float().contiguous()
    g = grad_out.detach().float().contiguous()
    act = active.contiguous()
    act = act.view(torch.uint8) if act.dtype == torch.bool else act.to(torch.uint8)
    lib = _lib.load()
    ws = torch.empty(max(int(lib.rdetr_relation_bias_backward_workspace_bytes(B, N1, N2)) // 4, 1), dtype=torch.float32, device=g.device)
    gw = torch.empty(Hh, 4 * num_pos_feats, dtype=torch.float32, device=g.device)
    gb = torch.empty(Hh, dtype=torch.float32, device=g.device)
    st = lib.rdetr_relation_bias_backward_f32(src.data_ptr(), tgt.data_ptr(), g.data_ptr(), act.data_ptr(), B, N1, N2, Hh, num_pos_feats,
                                              float(scale), float(temperature), float(eps), ws.data_ptr(), gw.data_ptr(), gb.data_ptr(),
                                              _stream_ptr(g))
    _lib.check(st, "rdetr_relation_bias_backward_f32")
    return gw, gb


def bias_softmax_(scores: torch.Tensor, bias: Optional[torch.Tensor] = None,
                  mask: Optional[torch.Tensor] = None) -> torch.Tensor:
    """In place: scores[BH,N1,N2] <- softmax(scores + bias, -1); mask [N1,N2] bool, True = excluded."""
    _require_device(scores, bias, mask)
    if scores.dtype != torch.float32 or scores.dim() != 3:
        raise _lib.RdetrError("scores must be a float32 [BH, N1, N2] tensor")
    _require_contiguous(scores=scores)
    BH, N1, N2 = scores.shape
    if bias is not None:
        if bias.dtype != torch.float32 or tuple(bias.shape) != (BH, N1, N2):
            raise _lib.RdetrError("bias must be float32 with the shape of scores")
        bias = bias.contiguous()
    mask_u8 = None
    if mask is not None:
        if tuple(mask.shape) != (N1, N2):
            raise _lib.RdetrError("mask must be [N1, N2]")
        mask_u8 = mask.to(torch.uint8).contiguous()
    st = _lib.load().rdetr_bias_softmax_f32(scores.data_ptr(), None if bias is None else bias.data_ptr(),
                                            None if mask_u8 is None else mask_u8.data_ptr(), BH, N1, N2,
                                            _stream_ptr(scores))
    _lib.check(st, "rdetr_bias_softmax_f32")
    return scores


def _row_matrix(t: torch.Tensor):
    """(tensor, row stride) if `t` is a [.., C] tensor whose rows are evenly strided with a contiguous last dim (a
    contiguous tensor or a column slice of one); else a contiguous copy."""
    C = t.shape[-1]
    if t.stride(-1) == 1 and t.dim() >= 1:
        ld = t.stride(-2) if t.dim() >= 2 else C
        ok = ld >= C
        for d in range(t.dim() - 3, -1, -1):                  # outer dims must continue the same row pitch
            ok = ok and (t.shape[d] == 1 or t.stride(d) == t.stride(d + 1) * t.shape[d + 1])
        if ok:
            return t, ld
    t = t.contiguous()
    return t, C


def add_layer_norm(x: torch.Tensor, residual: Optional[torch.Tensor], weight: torch.Tensor, bias: torch.Tensor,
                   eps: float = 1e-5, out: Optional[torch.Tensor] = None, pos: Optional[torch.Tensor] = None):
    """LayerNorm(x + residual) over the last dimension in one pass (fp32 / bf16; residual may be None).  `x`,
    `residual` and `out` may be column slices of wider row-major tensors (evenly strided rows).
    With ``pos`` (x's shape and dtype) returns ``(out, out + pos)`` -- the second tensor is what the next encoder layer
    feeds its attention as ``query + query_pos``, produced by the same pass.
    Inference-only (no autograd): the harness uses it under torch.no_grad()."""
    _require_device(x, residual, weight, bias, out)
    if x.dtype not in (torch.float32, torch.bfloat16):
        raise _lib.RdetrError(f"add_layer_norm: dtype {x.dtype} not supported (float32 or bfloat16)")
    C = x.shape[-1]
    if residual is not None and (residual.shape != x.shape or residual.dtype != x.dtype):
        raise _lib.RdetrError("add_layer_norm: residual must have x's shape and dtype")
    if weight.numel() != C or bias.numel() != C:
        raise _lib.RdetrError("add_layer_norm: weight / bias must have C elements")
    x, ldx = _row_matrix(x)
    ldr = C
    if residual is not None:
        residual, ldr = _row_matrix(residual)
    if out is None:
        out = torch.empty(x.shape, dtype=x.dtype, device=x.device)
    elif out.shape != x.shape or out.dtype != x.dtype:
        raise _lib.RdetrError("add_layer_norm: out must have x's shape and dtype")
    o, ldo = _row_matrix(out)
    if o is not out:
        raise _lib.RdetrError("add_layer_norm: out must have evenly strided rows with a contiguous last dimension")
    w, b = weight.detach().to(x.dtype).contiguous(), bias.detach().to(x.dtype).contiguous()
    lib = _lib.load()
    if pos is not None:
        _require_device(pos)
        if pos.shape != x.shape or pos.dtype != x.dtype:
            raise _lib.RdetrError("add_layer_norm: pos must have x's shape and dtype")
        pos, ldp = _row_matrix(pos)
        out2 = torch.empty(x.shape, dtype=x.dtype, device=x.device)
        fp = lib.rdetr_add_layernorm_pos_f32 if x.dtype == torch.float32 else lib.rdetr_add_layernorm_pos_bf16
        st = fp(x.data_ptr(), None if residual is None else residual.data_ptr(), w.data_ptr(), b.data_ptr(), pos.data_ptr(),
                x.numel() // C, C, ldx, ldr, ldo, ldp, C, float(eps), out.data_ptr(), out2.data_ptr(), _stream_ptr(x))
        _lib.check(st, "rdetr_add_layernorm_pos")
        return out, out2
    fn = lib.rdetr_add_layernorm_strided_f32 if x.dtype == torch.float32 else lib.rdetr_add_layernorm_strided_bf16
    st = fn(x.data_ptr(), None if residual is None else residual.data_ptr(), w.data_ptr(), b.data_ptr(),
            x.numel() // C, C, ldx, ldr, ldo, float(eps), out.data_ptr(), _stream_ptr(x))
    _lib.check(st, "rdetr_add_layernorm")
    return out


def relation_attention(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, num_heads: int, bias: Optional[torch.Tensor] = None,
                       mask: Optional[torch.Tensor] = None, scale: Optional[float] = None) -> torch.Tensor:
    """softmax(Q K^T * scale + bias) V per (image, head) in one kernel (bf16, head dim 32, inference).
    q [B,N,C], k / v [B,M,C] bf16 -- row-strided views are fine (last dim contiguous, image stride = rows * row stride);
    bias fp32 [B*H,N,M] or None; mask bool [N,M] (True = excluded) or None -> [B,N,C] bf16."""
    _require_device(q, k, v, bias, mask)
    if q.dtype != torch.bfloat16 or k.dtype != torch.bfloat16 or v.dtype != torch.bfloat16:
        raise _lib.RdetrError("relation_attention: q, k, v must be bfloat16")
    B, N, C = q.shape
    M = k.shape[1]
    if C % num_heads or k.shape != (B, M, C) or v.shape != (B, M, C):
        raise _lib.RdetrError("relation_attention: q [B,N,C], k / v [B,M,C] expected")
    D = C // num_heads

    def rows(t, n):
        if t.stride(2) != 1 or (t.shape[0] > 1 and t.stride(0) != n * t.stride(1)):
            t = t.contiguous()
        return t, t.stride(1)

    (q, ldq), (k, ldk), (v, ldv) = rows(q, N), rows(k, M), rows(v, M)
    if bias is not None:
        if bias.dtype != torch.float32 or bias.numel() != B * num_heads * N * M:
            raise _lib.RdetrError("relation_attention: bias must be float32 [B*H, N, M]")
        bias = bias.contiguous()
    mask_u8 = None
    if mask is not None:
        if tuple(mask.shape) != (N, M):
            raise _lib.RdetrError("relation_attention: mask must be [N, M]")
        mask_u8 = mask.to(torch.uint8).contiguous()
    out = torch.empty(B, N, C, dtype=torch.bfloat16, device=q.device)
    st = _lib.load().rdetr_relation_attention_bf16(
        q.data_ptr(), k.data_ptr(), v.data_ptr(), ldq, ldk, ldv, None if bias is None else bias.data_ptr(),
        None if mask_u8 is None else mask_u8.data_ptr(), B, num_heads, D, N, M,
        float(scale if scale is not None else D ** -0.5), out.data_ptr(), C, _stream_ptr(q))
    _lib.check(st, "rdetr_relation_attention_bf16")
    return out


def relation_attention_boxes(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, num_heads: int, src_boxes: torch.Tensor,
                             tgt_boxes: torch.Tensor, proj_weight: torch.Tensor, proj_bias: Optional[torch.Tensor],
                             mask: Optional[torch.Tensor] = None, scale: Optional[float] = None, num_pos_feats: int = 16,
                             temperature: float = 10000.0, rel_scale: float = 100.0, eps: float = 1e-5) -> torch.Tensor:
    """softmax(Q K^T * scale + relation_bias(src_boxes, tgt_boxes)) V with the bias generated inside the kernel (bf16, 8 heads of
    32, 16 sine features per coordinate, inference).  q [B,N,C], k / v [B,M,C] bf16 (row-strided views are fine); src_boxes
    [B,N,4] / tgt_boxes [B,M,4] cxcywh; proj_weight [8, 64(,1,1)], proj_bias [8]; mask bool [N,M] or None -> [B,N,C] bf16."""
    _require_device(q, k, v, src_boxes, tgt_boxes, proj_weight, proj_bias, mask)
    if q.dtype != torch.bfloat16 or k.dtype != torch.bfloat16 or v.dtype != torch.bfloat16:
        raise _lib.RdetrError("relation_attention_boxes: q, k, v must be bfloat16")
    B, N, C = q.shape
    M = k.shape[1]
    if C % num_heads or k.shape != (B, M, C) or v.shape != (B, M, C):
        raise _lib.RdetrError("relation_attention_boxes: q [B,N,C], k / v [B,M,C] expected")
    if tuple(src_boxes.shape) != (B, N, 4) or tuple(tgt_boxes.shape) != (B, M, 4):
        raise _lib.RdetrError("relation_attention_boxes: src_boxes [B,N,4] and tgt_boxes [B,M,4] expected")
    D = C // num_heads

    def rows(t, n):
        if t.stride(2) != 1 or (t.shape[0] > 1 and t.stride(0) != n * t.stride(1)):
            t = t.contiguous()
        return t, t.stride(1)

    (q, ldq), (k, ldk), (v, ldv) = rows(q, N), rows(k, M), rows(v, M)
    src = src_boxes.detach().float().contiguous()
    tgt = tgt_boxes.detach().float().contiguous()
    if proj_weight.numel() != num_heads * 4 * num_pos_feats:
        raise _lib.RdetrError(f"proj_weight must be [{num_heads}, {4 * num_pos_feats}]")
    # fp32 copies of the projection (the module holds it in the network dtype), kept until a parameter changes: converted per
    # call they were two small launches in every decoder layer's dependency chain
    def f32_copy(t, shape):
        def build():
            c = t.detach().float().reshape(shape).contiguous()
            if c.data_ptr() == t.data_ptr():
                c = c.clone()                           # an fp32 parameter: the cache must not alias it
            if not torch.cuda.is_current_stream_capturing():
                torch.cuda.current_stream(t.device).synchronize()      # other streams (image groups) pick it up without an event
            return c
        return _REL_PROJ_F32.get((t,), build)
    w = f32_copy(proj_weight, (num_heads, -1))
    pb = None if proj_bias is None else f32_copy(proj_bias, (-1,))
    mask_u8 = None
    if mask is not None:
        if tuple(mask.shape) != (N, M):
            raise _lib.RdetrError("relation_attention_boxes: mask must be [N, M]")
        mask_u8 = mask.to(torch.uint8).contiguous()
    out = torch.empty(B, N, C, dtype=torch.bfloat16, device=q.device)
    st = _lib.load().rdetr_relation_attention_boxes_bf16(
        q.data_ptr(), k.data_ptr(), v.data_ptr(), ldq, ldk, ldv, src.data_ptr(), tgt.data_ptr(), w.data_ptr(),
        None if pb is None else pb.data_ptr(), None if mask_u8 is None else mask_u8.data_ptr(), B, num_heads, D, N, M,
        num_pos_feats, float(rel_scale), float(temperature), float(eps), float(scale if scale is not None else D ** -0.5),
        out.data_ptr(), C, _stream_ptr(q))
    _lib.check(st, "rdetr_relation_attention_boxes_bf16")
    return out


def box_refine(delta: torch.Tensor, reference: torch.Tensor, eps: float = 1e-3) -> torch.Tensor:
    """sigmoid(delta + inverse_sigmoid(reference)) in one kernel; delta fp32 / bf16, reference fp32 -> fp32."""
    _require_device(delta, reference)
    if delta.shape != reference.shape or reference.dtype != torch.float32 or delta.dtype not in (torch.float32, torch.bfloat16):
        raise _lib.RdetrError("box_refine: delta (fp32 / bf16) and reference (fp32) must have the same shape")
    delta, reference = delta.contiguous(), reference.contiguous()
    out = torch.empty_like(reference)
    st = _lib.load().rdetr_box_refine_f32(delta.data_ptr(), int(delta.dtype == torch.bfloat16), reference.data_ptr(),
                                          reference.numel(), float(eps), out.data_ptr(), _stream_ptr(reference))
    _lib.check(st, "rdetr_box_refine_f32")
    return out


def sine_pos_embed(pos: torch.Tensor, num_pos_feats: int = 128, temperature: float = 10000.0, scale: float = 6.283185307179586,
                   dtype: torch.dtype = torch.float32) -> torch.Tensor:
    """get_sine_pos_embed(..., exchange_xy=True) in one kernel: pos fp32 [..., n] -> [..., n * num_pos_feats] (fp32 / bf16)."""
    _require_device(pos)
    if pos.dtype != torch.float32 or dtype not in (torch.float32, torch.bfloat16):
        raise _lib.RdetrError("sine_pos_embed: pos must be float32, output float32 or bfloat16")
    pos = pos.contiguous()
    n = pos.shape[-1]
    out = torch.empty(*pos.shape[:-1], n * num_pos_feats, dtype=dtype, device=pos.device)
    st = _lib.load().rdetr_sine_pos_embed(pos.data_ptr(), pos.numel() // n, n, num_pos_feats, float(temperature), float(scale),
                                          out.data_ptr(), int(dtype == torch.bfloat16), _stream_ptr(pos))
    _lib.check(st, "rdetr_sine_pos_embed")
    return out


def _rows_view(x: torch.Tensor, name: str):
    """(rows, C, row stride in elements) of a tensor whose last dimension is contiguous and whose leading dimensions
    collapse to evenly strided rows (a contiguous tensor or a column slice of one)."""
    if x.dim() < 2 or x.stride(-1) != 1:
        raise _lib.RdetrError(f"{name}: last dimension must be contiguous")
    C, ld = x.shape[-1], x.stride(-2)
    rows, expect = 1, ld
    for d in range(x.dim() - 2, -1, -1):
        if x.shape[d] != 1 and x.stride(d) != expect:
            raise _lib.RdetrError(f"{name}: rows must be evenly strided")
        expect *= x.shape[d]
        rows *= x.shape[d]
    return rows, C, ld


def zero_masked_rows_(x: torch.Tensor, mask: torch.Tensor) -> torch.Tensor:
    """In place ``x.masked_fill_(mask[..., None], 0)`` (ms_deform_attn.py:316-319) that writes only the masked rows.
    x [..., C] fp32|bf16 (rows evenly strided, 16-byte multiples), mask bool/uint8 with one entry per row."""
    _require_device(x, mask)
    rows, C, ld = _rows_view(x, "zero_masked_rows_")
    if mask.numel() != rows:
        raise _lib.RdetrError("zero_masked_rows_: mask needs one entry per row of x")
    m = mask.contiguous()
    m = m.view(torch.uint8) if m.dtype == torch.bool else m.to(torch.uint8)
    es = x.element_size()
    st = _lib.load().rdetr_zero_masked_rows(x.data_ptr(), m.data_ptr(), rows, C * es, ld * es, _stream_ptr(x))
    _lib.check(st, "rdetr_zero_masked_rows")
    return x


def row_max(x: torch.Tensor) -> torch.Tensor:
    """``x.max(-1)[0]`` for fp32|bf16 x [..., C] (relation_transformer.py:105); NaN propagates."""
    _require_device(x)
    if x.dtype not in (torch.float32, torch.bfloat16):
        raise _lib.RdetrError("row_max: float32 or bfloat16")
    rows, C, ld = _rows_view(x, "row_max")
    out = torch.empty(x.shape[:-1], dtype=x.dtype, device=x.device)
    st = _lib.load().rdetr_row_max(x.data_ptr(), int(x.dtype == torch.bfloat16), rows, C, ld, out.data_ptr(), _stream_ptr(x))
    _lib.check(st, "rdetr_row_max")
    return out


def _packed_k256(weight: torch.Tensor) -> torch.Tensor:
    """[256, 256] bf16 weight in MFMA-fragment order (rdetr_linear_pack_k256_bf16), cached until the tensor changes or dies."""
    def build():
        packed = torch.empty(256 * 256, dtype=torch.bfloat16, device=weight.device)
        st = _lib.load().rdetr_linear_pack_k256_bf16(weight.data_ptr(), packed.data_ptr(), _stream_ptr(weight))
        _lib.check(st, "rdetr_linear_pack_k256_bf16")
        if not torch.cuda.is_current_stream_capturing():
            torch.cuda.current_stream(weight.device).synchronize()
        return packed
    return _LINEAR_PACKED.get((weight,), build)


def box_head_k256_supported(x: torch.Tensor, layers) -> bool:
    return (x.is_cuda and x.dtype == torch.bfloat16 and x.shape[-1] == 256 and len(layers) == 3
            and tuple(layers[0].weight.shape) == (256, 256) and tuple(layers[1].weight.shape) == (256, 256)
            and tuple(layers[2].weight.shape) == (4, 256) and all(l.bias is not None and l.weight.dtype == torch.bfloat16
                                                                  and l.weight.is_contiguous() and l.weight.data_ptr() % 16 == 0 for l in layers))


def box_head_k256(xa: torch.Tensor, xb: Optional[torch.Tensor], layers, reference: torch.Tensor, eps: float = 1e-3,
                  reference_is_logit: bool = False):
    """``sigmoid(MLP3(x) + inverse_sigmoid(reference))`` for the decoder's box head (three nn.Linear: 256 -> 256 -> 256 -> 4, ReLU
    between) on one or two bf16 [..., 256] inputs sharing the fp32 reference boxes [..., 4]: one kernel instead of 3 (6) GEMMs
    and 1 (2) refine launches (relation_transformer.py:294, 363-381).  ``reference_is_logit``: the reference is added as it is
    (the two-stage proposals, :88-90).  Returns fp32 boxes (a pair when ``xb`` is given)."""
    _require_device(xa, xb, reference)
    if not box_head_k256_supported(xa, layers) or reference.dtype != torch.float32 or reference.shape[-1] != 4:
        raise _lib.RdetrError("box_head_k256: bf16 [..., 256] inputs, Linear(256,256), Linear(256,256), Linear(256,4) in bf16, fp32 reference")
    rows, _, lda = _rows_view(xa, "box_head_k256")
    ldb = 0
    if xb is not None:
        if xb.shape != xa.shape or xb.dtype != xa.dtype:
            raise _lib.RdetrError("box_head_k256: the two inputs must have one shape and dtype")
        _, _, ldb = _rows_view(xb, "box_head_k256")
    ref = reference.contiguous()
    if ref.numel() != rows * 4:
        raise _lib.RdetrError("box_head_k256: reference must hold one box per input row")
    pw1, pw2 = _packed_k256(layers[0].weight), _packed_k256(layers[1].weight)
    out_a = torch.empty_like(ref)
    out_b = torch.empty_like(ref) if xb is not None else None
    st = _lib.load().rdetr_box_head_k256_bf16(
        xa.data_ptr(), lda, None if xb is None else xb.data_ptr(), ldb, pw1.data_ptr(), _cptr(layers[0].bias),
        pw2.data_ptr(), _cptr(layers[1].bias), layers[2].weight.data_ptr(), _cptr(layers[2].bias),
        ref.data_ptr(), int(reference_is_logit), float(eps), rows, out_a.data_ptr(), None if out_b is None else out_b.data_ptr(),
        _stream_ptr(xa))
    _lib.check(st, "rdetr_box_head_k256_bf16")
    return out_a if xb is None else (out_a, out_b)


def query_pos_k256_supported(emb: torch.Tensor, query: torch.Tensor, head_layers, scale_layers=None) -> bool:
    def lin_ok(l, n_in):
        return (tuple(l.weight.shape) == (256, n_in) and l.bias is not None and l.weight.dtype == torch.bfloat16
                and l.bias.dtype == torch.bfloat16 and l.weight.is_contiguous() and l.weight.data_ptr() % 16 == 0)
    if not (emb.is_cuda and emb.dtype == torch.bfloat16 and query.dtype == torch.bfloat16 and emb.shape[-1] == 512 and query.shape[-1] == 256
            and emb.shape[:-1] == query.shape[:-1] and len(head_layers) == 2 and lin_ok(head_layers[0], 512) and lin_ok(head_layers[1], 256)):
        return False
    return scale_layers is None or (len(scale_layers) == 2 and lin_ok(scale_layers[0], 256) and lin_ok(scale_layers[1], 256))


def _packed_k_halves(weight: torch.Tensor):
    """The two K halves of a [256, 512] bf16 weight, each in fragment order; cached until the tensor changes or dies."""
    def build():
        halves = []
        for h in range(2):
            w = weight.detach()[:, 256 * h:256 * (h + 1)].contiguous()
            packed = torch.empty(256 * 256, dtype=torch.bfloat16, device=weight.device)
            st = _lib.load().rdetr_linear_pack_k256_bf16(w.data_ptr(), packed.data_ptr(), _stream_ptr(weight))
            _lib.check(st, "rdetr_linear_pack_k256_bf16")
            halves.append(packed)
        if not torch.cuda.is_current_stream_capturing():
            torch.cuda.current_stream(weight.device).synchronize()
        return tuple(halves)
    return _LINEAR_PACKED.get((weight,), build)


def query_pos_k256(emb: torch.Tensor, query: torch.Tensor, head_layers, scale_layers=None):
    """``(query_pos, query + query_pos)`` of a decoder layer in one kernel (csrc/qpos.hip): ``query_pos = ref_point_head(emb)``,
    multiplied by ``query_scale(query)`` when ``scale_layers`` is given (relation_transformer.py:343-347).  emb [..., 512], query
    [..., 256] bf16; head_layers / scale_layers: the two nn.Linear of each MLP.  Inference only."""
    _require_device(emb, query)
    if not query_pos_k256_supported(emb, query, head_layers, scale_layers):
        raise _lib.RdetrError("query_pos_k256: bf16 emb [..., 512] / query [..., 256], Linear(512,256), Linear(256,256) (x2) in bf16 with biases")
    if query.dim() == 3 and query.stride(0) == 0:
        query = query.contiguous()                      # layer 0: tgt_embed.weight.expand(B, -1, -1) (relation_transformer.py:117)
    rows, _, lde = _rows_view(emb, "query_pos_k256")
    _, _, ldq = _rows_view(query, "query_pos_k256")
    if lde % 8 or ldq % 8 or emb.data_ptr() % 16 or query.data_ptr() % 16:
        raise _lib.RdetrError("query_pos_k256: rows must be 16-byte aligned")
    p1a, p1b = _packed_k_halves(head_layers[0].weight)
    p2 = _packed_k256(head_layers[1].weight)
    sc = [None] * 4
    if scale_layers is not None:
        sc = [_packed_k256(scale_layers[0].weight).data_ptr(), _cptr(scale_layers[0].bias),
              _packed_k256(scale_layers[1].weight).data_ptr(), _cptr(scale_layers[1].bias)]
    pos = torch.empty(*query.shape, dtype=torch.bfloat16, device=query.device)
    qpp = torch.empty_like(pos)
    st = _lib.load().rdetr_query_pos_k256_bf16(emb.data_ptr(), lde, query.data_ptr(), ldq, p1a.data_ptr(), p1b.data_ptr(),
                                               _cptr(head_layers[0].bias), p2.data_ptr(),
                                               _cptr(head_layers[1].bias), sc[0], sc[1], sc[2], sc[3], rows,
                                               pos.data_ptr(), qpp.data_ptr(), _stream_ptr(query))
    _lib.check(st, "rdetr_query_pos_k256_bf16")
    return pos, qpp


def encoder_proj_supported(x: torch.Tensor, xq: torch.Tensor, wv: torch.Tensor, wq: torch.Tensor) -> bool:
    if not (x.is_cuda and x.dim() == 3 and xq.shape == x.shape and x.dtype == torch.bfloat16 and xq.dtype == torch.bfloat16
            and x.shape[-1] == 256 and wv.dtype == torch.bfloat16 and wq.dtype == torch.bfloat16 and tuple(wv.shape) == (256, 256)
            and tuple(wq.shape) in ((384, 256), (480, 256)) and wv.is_contiguous() and wq.is_contiguous() and wv.data_ptr() % 16 == 0):
        return False
    try:
        for t in (x, xq):
            _, _, ld = _rows_view(t, "encoder_proj")
            if ld % 8 or t.data_ptr() % 16:
                return False
    except _lib.RdetrError:
        return False
    return True


def _packed_query_proj(wq: torch.Tensor) -> torch.Tensor:
    """[384 | 480, 256] merged query-projection weight -> two packed [256, 256] blocks (zero rows up to 512); cached."""
    def build():
        full = torch.zeros(512, 256, dtype=torch.bfloat16, device=wq.device)
        full[:wq.shape[0]].copy_(wq.detach())
        packed = torch.empty(2 * 256 * 256, dtype=torch.bfloat16, device=wq.device)
        for blk in range(2):
            st = _lib.load().rdetr_linear_pack_k256_bf16(full[256 * blk:256 * (blk + 1)].data_ptr(), packed.data_ptr() + blk * 256 * 256 * 2,
                                                         _stream_ptr(wq))
            _lib.check(st, "rdetr_linear_pack_k256_bf16")
        if not torch.cuda.is_current_stream_capturing():
            torch.cuda.current_stream(wq.device).synchronize()
        return packed
    return _LINEAR_PACKED.get((wq,), build)


def encoder_proj(x: torch.Tensor, xq: torch.Tensor, wv: torch.Tensor, bv: Optional[torch.Tensor], wq: torch.Tensor,
                 bq: Optional[torch.Tensor], key_padding_mask: Optional[torch.Tensor] = None):
    """The three input projections of an encoder layer's MSDA in one kernel (csrc/proj.hip): x, xq [B, S, 256] bf16 (rows may be
    column slices); wv [256, 256] = value_proj.weight, wq [384 | 480, 256] = [sampling_offsets.weight ; attention_weights.weight]
    (4 | 5 feature levels) -> (value head-major [B, 8, S, 32] with padded rows zero, raw [B, S, 384 | 480] offsets | logits).
    Inference only."""
    _require_device(x, xq, wv, bv, wq, bq, key_padding_mask)
    if not encoder_proj_supported(x, xq, wv, wq):
        raise _lib.RdetrError("encoder_proj: bf16 [B, S, 256] inputs with 16-byte aligned rows, wv [256, 256], wq [384 | 480, 256]")
    B, S, _ = x.shape
    q_cols = wq.shape[0]
    _, _, ldx = _rows_view(x, "encoder_proj")
    _, _, ldq = _rows_view(xq, "encoder_proj")
    for t, n in ((bv, 256), (bq, q_cols)):
        if t is not None and (t.dtype != torch.bfloat16 or t.numel() != n):
            raise _lib.RdetrError("encoder_proj: biases must be bf16 [256] / [384 | 480]")
    mask_ptr = None
    if key_padding_mask is not None:
        if tuple(key_padding_mask.shape) != (B, S):
            raise _lib.RdetrError("key_padding_mask must be [B, S]")
        mask_u8 = key_padding_mask.contiguous().view(torch.uint8) if key_padding_mask.dtype == torch.bool \
            else key_padding_mask.to(torch.uint8).contiguous()
        mask_ptr = mask_u8.data_ptr()
    out_hm = torch.empty(B, 8, S, 32, dtype=torch.bfloat16, device=x.device)
    out_q = torch.empty(B, S, q_cols, dtype=torch.bfloat16, device=x.device)
    st = _lib.load().rdetr_encoder_proj_k256_bf16(
        x.data_ptr(), ldx, xq.data_ptr(), ldq, _packed_k256(wv).data_ptr(), None if bv is None else _cptr(bv),
        _packed_query_proj(wq).data_ptr(), None if bq is None else _cptr(bq), mask_ptr, B, S, q_cols, out_hm.data_ptr(),
        out_q.data_ptr(), _stream_ptr(x))
    _lib.check(st, "rdetr_encoder_proj_k256_bf16")
    return out_hm, out_q


def topk_supported(x: torch.Tensor, k: int) -> bool:
    return (x.is_cuda and x.dim() == 2 and x.dtype in (torch.float32, torch.bfloat16) and 1 <= k <= min(x.shape[1], 1024)
            and x.shape[1] < (1 << 20) and x.shape[0] <= 65535)


def topk(x: torch.Tensor, k: int):
    """``torch.topk(x, k, dim=1)`` (largest, sorted) for fp32 | bf16 x [rows, n]: (values fp32 [rows, k], indices int64 [rows, k]);
    equal values come out by ascending index, NaN first (csrc/topk.hip).  The two selections on the transformer's chain
    (relation_transformer.py:93, post_process.py:30)."""
    _require_device(x)
    if not topk_supported(x, k):
        raise _lib.RdetrError("topk: fp32 | bf16 [rows, n] on the device, 1 <= k <= min(n, 1024), n < 2^20")
    x = x.contiguous()
    rows, n = x.shape
    lib = _lib.load()
    ws = torch.empty(int(lib.rdetr_topk_workspace_bytes(rows, n, k)) + 16, dtype=torch.uint8, device=x.device)
    values = torch.empty(rows, k, dtype=torch.float32, device=x.device)
    indices = torch.empty(rows, k, dtype=torch.int64, device=x.device)
    off = (-ws.data_ptr()) % 16
    st = lib.rdetr_topk(x.data_ptr(), int(x.dtype == torch.bfloat16), rows, n, k, ws.data_ptr() + off, values.data_ptr(),
                        indices.data_ptr(), _stream_ptr(x))
    _lib.check(st, "rdetr_topk")
    return values, indices


def detections_from_topk(score: torch.Tensor, index: torch.Tensor, boxes: torch.Tensor, image_sizes: torch.Tensor, num_classes: int):
    """[B,K,6] = (x1, y1, x2, y2, score, label) from PostProcess's top-k (post_process.py:30-44): score fp32 [B,K], index int64 [B,K]
    into the flattened [N * C] scores, boxes fp32 [B,N,4] cxcywh, image_sizes int64 [B,2] (h, w)."""
    _require_device(score, index, boxes, image_sizes)
    B, K = score.shape
    if (score.dtype != torch.float32 or index.dtype != torch.int64 or boxes.dtype != torch.float32 or image_sizes.dtype != torch.int64
            or tuple(index.shape) != (B, K) or boxes.dim() != 3 or boxes.shape[0] != B or boxes.shape[2] != 4 or tuple(image_sizes.shape) != (B, 2)):
        raise _lib.RdetrError("detections_from_topk: score fp32 [B,K], index int64 [B,K], boxes fp32 [B,N,4], image_sizes int64 [B,2]")
    score, index, boxes, image_sizes = score.contiguous(), index.contiguous(), boxes.contiguous(), image_sizes.contiguous()
    out = torch.empty(B, K, 6, dtype=torch.float32, device=score.device)
    st = _lib.load().rdetr_detections_from_topk(score.data_ptr(), index.data_ptr(), boxes.data_ptr(), image_sizes.data_ptr(), B,
                                                boxes.shape[1], num_classes, K, out.data_ptr(), _stream_ptr(score))
    _lib.check(st, "rdetr_detections_from_topk")
    return out


def scaled_pos(a: torch.Tensor, scale: torch.Tensor, query: torch.Tensor):
    """``(a * scale, query + a * scale)`` in one pass (same rounding as the two torch kernels); same-shape fp32 | bf16 tensors."""
    _require_device(a, scale, query)
    if a.dtype not in (torch.float32, torch.bfloat16) or scale.dtype != a.dtype or query.dtype != a.dtype or scale.shape != a.shape \
            or query.shape != a.shape:
        raise _lib.RdetrError("scaled_pos: three tensors of one shape and dtype (float32 or bfloat16)")
    a, scale, query = a.contiguous(), scale.contiguous(), query.contiguous()
    pos, qp = torch.empty_like(a), torch.empty_like(a)
    st = _lib.load().rdetr_scaled_pos(a.data_ptr(), scale.data_ptr(), query.data_ptr(), a.numel(), int(a.dtype == torch.bfloat16),
                                      pos.data_ptr(), qp.data_ptr(), _stream_ptr(a))
    _lib.check(st, "rdetr_scaled_pos")
    return pos, qp


def decoder_reference(reference_points: torch.Tensor, valid_ratios: torch.Tensor, num_pos_feats: int = 128, temperature: float = 10000.0,
                      scale: float = 6.283185307179586, dtype: torch.dtype = torch.float32):
    """(ref_in [B,N,L,4] fp32, sine embedding of ref_in[:, :, 0, :] [B,N,4*num_pos_feats] in ``dtype``) from the decoder's reference
    boxes [B,N,4] fp32 and the valid ratios [B,L,2] fp32: relation_transformer.py:335-343 in one launch."""
    _require_device(reference_points, valid_ratios)
    if (reference_points.dtype != torch.float32 or valid_ratios.dtype != torch.float32 or reference_points.dim() != 3
            or reference_points.shape[-1] != 4 or valid_ratios.dim() != 3 or valid_ratios.shape[-1] != 2
            or valid_ratios.shape[0] != reference_points.shape[0] or dtype not in (torch.float32, torch.bfloat16)):
        raise _lib.RdetrError("decoder_reference: reference [B,N,4] fp32, valid_ratios [B,L,2] fp32")
    ref, vr = reference_points.contiguous(), valid_ratios.contiguous()
    B, N, _ = ref.shape
    L = vr.shape[1]
    ref_in = torch.empty(B, N, L, 4, dtype=torch.float32, device=ref.device)
    emb = torch.empty(B, N, 4 * num_pos_feats, dtype=dtype, device=ref.device)
    st = _lib.load().rdetr_decoder_reference(ref.data_ptr(), vr.data_ptr(), B, N, L, num_pos_feats, float(temperature), float(scale),
                                             ref_in.data_ptr(), emb.data_ptr(), int(dtype == torch.bfloat16), _stream_ptr(ref))
    _lib.check(st, "rdetr_decoder_reference")
    return ref_in, emb


def pyramid_points(level_masks, pad_mask: Optional[torch.Tensor], keep_dtype: torch.dtype):
    """(valid_ratios [B,L,2], reference points [B,S,L,2], proposal logits [B,S,4], keep [B,S]) of a padded pyramid in two
    launches (base_transformer.py:42-70, relation_transformer.py:162-176).  level_masks: bool [B,h_l,w_l] per level; pad_mask:
    their flattening [B,S] or None; keep_dtype float32 | bfloat16."""
    import ctypes
    _require_device(*level_masks, pad_mask)
    L, B = len(level_masks), level_masks[0].shape[0]
    masks = [m.contiguous() for m in level_masks]
    if any(m.dtype != torch.bool or m.dim() != 3 or m.shape[0] != B for m in masks) or keep_dtype not in (torch.float32, torch.bfloat16):
        raise _lib.RdetrError("pyramid_points: bool masks [B, h, w] per level, keep in float32 or bfloat16")
    S = sum(m.shape[1] * m.shape[2] for m in masks)
    dev = masks[0].device
    pad = None
    if pad_mask is not None:
        if tuple(pad_mask.shape) != (B, S) or pad_mask.dtype != torch.bool:
            raise _lib.RdetrError("pyramid_points: pad_mask must be bool [B, S]")
        pad = pad_mask.contiguous()
    ptrs = (ctypes.c_void_p * L)(*[m.data_ptr() for m in masks])
    hw = (ctypes.c_int * (2 * L))(*[v for m in masks for v in (m.shape[1], m.shape[2])])
    ratios = torch.empty(B, L, 2, dtype=torch.float32, device=dev)
    reference = torch.empty(B, S, L, 2, dtype=torch.float32, device=dev)
    logit = torch.empty(B, S, 4, dtype=torch.float32, device=dev)
    keep = torch.empty(B, S, dtype=keep_dtype, device=dev)
    st = _lib.load().rdetr_pyramid_points(ptrs, hw, L, B, None if pad is None else pad.data_ptr(), int(keep_dtype == torch.bfloat16),
                                          ratios.data_ptr(), reference.data_ptr(), logit.data_ptr(), keep.data_ptr(), _stream_ptr(ratios))
    _lib.check(st, "rdetr_pyramid_points")
    return ratios, reference, logit, keep


def tokens_from_levels(levels, add_vecs=None, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """Level-packed tokens of a feature pyramid: ``cat([x.flatten(2).transpose(1, 2) (+ add_vecs[l]) for x in levels], 1)``
    (base_transformer.py:17-23; the optional per-channel vectors are the level embeddings, relation_transformer.py:87-89).
    levels: [B, C, h_l, w_l] fp32|bf16 contiguous; out: optional [B, S, C] destination whose rows may be a column slice of a
    wider buffer."""
    x0 = levels[0]
    _require_device(*levels)
    B, C = x0.shape[:2]
    S = sum(x.shape[2] * x.shape[3] for x in levels)
    if out is None:
        out = torch.empty(B, S, C, dtype=x0.dtype, device=x0.device)
    if tuple(out.shape) != (B, S, C) or out.dtype != x0.dtype or out.stride(-1) != 1 or (B > 1 and out.stride(0) < S * out.stride(1)):
        raise _lib.RdetrError("tokens_from_levels: out must be [B, S, C] in the levels' dtype with contiguous channels")
    lib, es, row = _lib.load(), out.element_size(), 0
    for l, x in enumerate(levels):
        if x.shape[:2] != (B, C) or x.dtype != x0.dtype:
            raise _lib.RdetrError("tokens_from_levels: levels must agree in batch, channels and dtype")
        x = x.contiguous()
        vec = None
        if add_vecs is not None:
            vec = add_vecs[l].to(x0.dtype).contiguous()
            if vec.numel() != C:
                raise _lib.RdetrError("tokens_from_levels: add_vecs[l] needs C entries")
        P = x.shape[2] * x.shape[3]
        st = lib.rdetr_nchw_to_tokens(x.data_ptr(), vec.data_ptr() if vec is not None else None, int(x0.dtype == torch.bfloat16),
                                      B, C, P, out.stride(0), out.stride(1), out.data_ptr() + row * out.stride(1) * es,
                                      _stream_ptr(x0))
        _lib.check(st, "rdetr_nchw_to_tokens")
        row += P
    return out


def linear_k256_supported(x: torch.Tensor, weight: torch.Tensor) -> bool:
    """True where `linear_k256` applies: bf16 device tensors, K = 256, N a multiple of 32, evenly strided 16-byte aligned rows."""
    if not (x.is_cuda and x.dtype == torch.bfloat16 and weight.dtype == torch.bfloat16 and x.shape[-1] == 256
            and weight.dim() == 2 and weight.shape[1] == 256 and weight.shape[0] % 32 == 0 and weight.is_contiguous()):
        return False
    try:
        _, _, ld = _rows_view(x, "linear_k256")
    except _lib.RdetrError:
        return False
    return ld % 8 == 0 and x.data_ptr() % 16 == 0 and weight.data_ptr() % 16 == 0


def linear_k256(x: torch.Tensor, weight: torch.Tensor, bias: Optional[torch.Tensor] = None, relu: bool = False,
                out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """``relu?(F.linear(x, weight, bias))`` for bf16 x [..., 256] (rows may be a column slice of a wider buffer) and
    weight [N, 256], N % 32 == 0, through the hand-written MFMA kernel (csrc/linear.hip).  Inference only."""
    _require_device(x, weight, bias, out)
    if not linear_k256_supported(x, weight):
        raise _lib.RdetrError("linear_k256: needs bf16, K = 256, N % 32 == 0, evenly strided 16-byte aligned rows")
    rows, _, ldx = _rows_view(x, "linear_k256")
    N = weight.shape[0]
    if bias is not None and (bias.dtype != torch.bfloat16 or bias.numel() != N):
        raise _lib.RdetrError("linear_k256: bias must be bf16 [N]")
    if out is None:
        out = torch.empty(*x.shape[:-1], N, dtype=x.dtype, device=x.device)
    elif tuple(out.shape) != (*x.shape[:-1], N) or out.dtype != x.dtype:
        raise _lib.RdetrError("linear_k256: out must be [..., N] bf16")
    orows, _, ldo = _rows_view(out, "linear_k256")
    if orows != rows or ldo % 8 or out.data_ptr() % 16:
        raise _lib.RdetrError("linear_k256: out rows must be 16-byte aligned")
    st = _lib.load().rdetr_linear_k256_bf16(x.data_ptr(), ldx, weight.data_ptr(), None if bias is None else _cptr(bias),
                                            rows, N, int(relu), out.data_ptr(), ldo, _stream_ptr(x))
    _lib.check(st, "rdetr_linear_k256_bf16")
    return out


def value_proj_head_major(x: torch.Tensor, weight: torch.Tensor, bias: Optional[torch.Tensor],
                          key_padding_mask: Optional[torch.Tensor] = None) -> torch.Tensor:
    """MSDA's value projection + padding zero-fill (ms_deform_attn.py:316-321) written straight into the head-major layout:
    x [B, S, 256] bf16 (rows may be a column slice of a wider buffer), weight [256, 256], bias [256] -> [B, 8, S, 32] bf16, rows of
    padded positions zero.  One MFMA kernel (csrc/linear.hip); same bits as F.linear + value_to_head_major."""
    _require_device(x, weight, bias, key_padding_mask)
    if x.dim() != 3 or not linear_k256_supported(x, weight) or tuple(weight.shape) != (256, 256):
        raise _lib.RdetrError("value_proj_head_major: needs bf16 x [B, S, 256] with evenly strided 16-byte aligned rows and a [256, 256] weight")
    B, S, _ = x.shape
    _, _, ldx = _rows_view(x, "value_proj_head_major")
    if bias is not None and (bias.dtype != torch.bfloat16 or bias.numel() != 256):
        raise _lib.RdetrError("value_proj_head_major: bias must be bf16 [256]")
    mask_ptr = None
    if key_padding_mask is not None:
        if tuple(key_padding_mask.shape) != (B, S):
            raise _lib.RdetrError("key_padding_mask must be [B, S]")
        mask_u8 = key_padding_mask.contiguous().view(torch.uint8) if key_padding_mask.dtype == torch.bool \
            else key_padding_mask.to(torch.uint8).contiguous()
        mask_ptr = mask_u8.data_ptr()
    out = torch.empty(B, 8, S, 32, dtype=torch.bfloat16, device=x.device)
    st = _lib.load().rdetr_linear_k256_hm_bf16(x.data_ptr(), ldx, weight.data_ptr(), None if bias is None else _cptr(bias),
                                               mask_ptr, B, S, out.data_ptr(), _stream_ptr(x))
    _lib.check(st, "rdetr_linear_k256_hm_bf16")
    return out


def ffn_k256_supported(x: torch.Tensor, w1: torch.Tensor, w2: torch.Tensor) -> bool:
    F = w1.shape[0] if w1.dim() == 2 else 0
    return (linear_k256_supported(x, w1) and w2.dtype == torch.bfloat16 and tuple(w2.shape) == (256, F) and F % 64 == 0
            and F <= 4096 and w2.is_contiguous() and w2.data_ptr() % 16 == 0)


class _PackedWeightCache:
    """Packed (fragment-order) copies of weight tensors, tied to the tensor OBJECTS: an entry holds weak references to its
    source tensors and is only a hit while every reference still resolves to the very tensor passed in and the versions
    match.  A freed model whose storage address is recycled by the caching allocator for another model's weights (same
    `data_ptr()`, same `_version` after an identical build sequence) can therefore never alias a stale packed copy --
    the pattern `host_levels` uses for the shape tables.
    Lifetime: a packed copy lives exactly as long as its source tensors.  The cache holds the ONLY reference to a copy whose
    address a captured HIP graph may have baked in (relation_detr_amd/graph.py replays kernels that read it), so an entry is
    never evicted while its sources are alive -- only entries whose sources are gone are pruned (every `prune_every` misses);
    the dictionary simply grows with the number of live weight tensors (a size cap with `clear()` freed copies
    that live graphs of a second network still read)."""

    def __init__(self, prune_every: int = 64):
        self._entries: dict = {}
        self._prune_every = prune_every
        self._misses = 0

    def get(self, tensors, build):
        key = tuple(id(t) for t in tensors)
        ver = tuple((t._version, tuple(t.shape), t.data_ptr(), t.device) for t in tensors)
        hit = self._entries.get(key)
        if hit is not None:
            refs, hver, packed = hit
            if hver == ver and all(r() is t for r, t in zip(refs, tensors)):
                return packed
        self._misses += 1
        if self._misses % self._prune_every == 0:       # drop the entries whose tensors are gone; live ones are never evicted
            self._entries = {k: v for k, v in self._entries.items() if all(r() is not None for r in v[0])}
        packed = build()
        self._entries[key] = (tuple(weakref.ref(t) for t in tensors), ver, packed)
        return packed

    def __len__(self):
        return len(self._entries)


_FFN_PACKED = _PackedWeightCache()        # (w1, w2) -> the fragment-order copy of a layer's two weight matrices


def ffn_k256_packed_weights(w1: torch.Tensor, w2: torch.Tensor) -> torch.Tensor:
    """The (w1, w2) pair re-ordered for `ffn_k256` (rdetr_ffn_k256_pack_bf16); cached until either tensor changes or dies."""
    def build():
        F = w1.shape[0]
        packed = torch.empty(2 * 256 * F, dtype=torch.bfloat16, device=w1.device)
        st = _lib.load().rdetr_ffn_k256_pack_bf16(w1.data_ptr(), w2.data_ptr(), F, packed.data_ptr(), _stream_ptr(w1))
        _lib.check(st, "rdetr_ffn_k256_pack_bf16")
        if not torch.cuda.is_current_stream_capturing():
            # once per weight update: other streams (image groups) pick the cached tensor up without an event of their own
            torch.cuda.current_stream(w1.device).synchronize()
        return packed
    return _FFN_PACKED.get((w1, w2), build)


def ffn_k256(x: torch.Tensor, w1: torch.Tensor, b1: torch.Tensor, w2: torch.Tensor, b2: torch.Tensor,
             out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """``F.linear(F.relu(F.linear(x, w1, b1)), w2, b2)`` for bf16 x [..., 256], w1 [F, 256], w2 [256, F] in one kernel: the
    [.., F] activations stay in registers (csrc/ffn.hip; relation_transformer.py:226-233, 272-275).  Inference only."""
    _require_device(x, w1, b1, w2, b2, out)
    if not ffn_k256_supported(x, w1, w2):
        raise _lib.RdetrError("ffn_k256: needs bf16, embed_dim 256, d_ffn % 64 == 0 (<= 4096), evenly strided 16-byte aligned rows")
    F = w1.shape[0]
    if b1.dtype != torch.bfloat16 or b2.dtype != torch.bfloat16 or b1.numel() != F or b2.numel() != 256:
        raise _lib.RdetrError("ffn_k256: biases must be bf16 [F] and [256]")
    rows, _, ldx = _rows_view(x, "ffn_k256")
    if out is None:
        out = torch.empty(*x.shape[:-1], 256, dtype=x.dtype, device=x.device)
    elif tuple(out.shape) != tuple(x.shape) or out.dtype != x.dtype:
        raise _lib.RdetrError("ffn_k256: out must have x's shape and dtype")
    orows, _, ldo = _rows_view(out, "ffn_k256")
    if orows != rows or ldo % 8 or out.data_ptr() % 16:
        raise _lib.RdetrError("ffn_k256: out rows must be 16-byte aligned")
    packed = ffn_k256_packed_weights(w1, w2)
    st = _lib.load().rdetr_ffn_k256_bf16(x.data_ptr(), ldx, packed.data_ptr(), _cptr(b1),
                                         _cptr(b2), rows, F, out.data_ptr(), ldo, _stream_ptr(x))
    _lib.check(st, "rdetr_ffn_k256_bf16")
    return out


def ffn_ln_k256(x: torch.Tensor, w1: torch.Tensor, b1: torch.Tensor, w2: torch.Tensor, b2: torch.Tensor, gamma: torch.Tensor,
                beta: torch.Tensor, eps: float = 1e-5, out: Optional[torch.Tensor] = None, pos: Optional[torch.Tensor] = None):
    """``LayerNorm(x + linear2(relu(linear1(x))))`` in one kernel (the end of an encoder / decoder layer,
    relation_transformer.py:272-276); with ``pos`` returns ``(out, out + pos)``.  bf16, inference only; see `ffn_k256`."""
    _require_device(x, w1, b1, w2, b2, gamma, beta, out, pos)
    if not ffn_k256_supported(x, w1, w2):
        raise _lib.RdetrError("ffn_ln_k256: needs bf16, embed_dim 256, d_ffn % 64 == 0 (<= 4096), evenly strided 16-byte aligned rows")
    F = w1.shape[0]
    for t, n in ((b1, F), (b2, 256), (gamma, 256), (beta, 256)):
        if t.dtype != torch.bfloat16 or t.numel() != n:
            raise _lib.RdetrError("ffn_ln_k256: biases / LayerNorm parameters must be bf16 of the matching size")
    rows, _, ldx = _rows_view(x, "ffn_ln_k256")
    if out is None:
        out = torch.empty(*x.shape[:-1], 256, dtype=x.dtype, device=x.device)
    elif tuple(out.shape) != tuple(x.shape) or out.dtype != x.dtype:
        raise _lib.RdetrError("ffn_ln_k256: out must have x's shape and dtype")
    orows, _, ldo = _rows_view(out, "ffn_ln_k256")
    if orows != rows or ldo % 8 or out.data_ptr() % 16:
        raise _lib.RdetrError("ffn_ln_k256: out rows must be 16-byte aligned")
    out2, ldp = None, 0
    if pos is not None:
        if tuple(pos.shape) != tuple(x.shape) or pos.dtype != x.dtype:
            raise _lib.RdetrError("ffn_ln_k256: pos must have x's shape and dtype")
        prows, _, ldp = _rows_view(pos, "ffn_ln_k256")
        if ldp % 8 or pos.data_ptr() % 16:
            raise _lib.RdetrError("ffn_ln_k256: pos rows must be 16-byte aligned")
        out2 = torch.empty(*x.shape[:-1], 256, dtype=x.dtype, device=x.device)
    packed = ffn_k256_packed_weights(w1, w2)
    st = _lib.load().rdetr_ffn_ln_k256_bf16(x.data_ptr(), ldx, packed.data_ptr(), _cptr(b1), _cptr(b2),
                                            _cptr(gamma), _cptr(beta), float(eps),
                                            None if pos is None else pos.data_ptr(), ldp, rows, F, out.data_ptr(), ldo,
                                            None if out2 is None else out2.data_ptr(), 256, _stream_ptr(x))
    _lib.check(st, "rdetr_ffn_ln_k256_bf16")
    return out if pos is None else (out, out2)


_LINEAR_PACKED = _PackedWeightCache()     # (weight,) -> fragment-order copy
_REL_PROJ_F32 = _PackedWeightCache()      # (pos_proj weight | bias,) -> fp32 copy


def linear_ln_k256_supported(x: torch.Tensor, weight: torch.Tensor, residual: torch.Tensor) -> bool:
    if not (linear_k256_supported(x, weight) and tuple(weight.shape) == (256, 256) and residual.dtype == torch.bfloat16
            and residual.is_cuda and tuple(residual.shape) == tuple(x.shape)):
        return False
    try:
        _, _, ldr = _rows_view(residual, "linear_ln_k256")
    except _lib.RdetrError:
        return False
    return ldr % 8 == 0 and residual.data_ptr() % 16 == 0


def linear_ln_k256(x: torch.Tensor, weight: torch.Tensor, bias: Optional[torch.Tensor], residual: torch.Tensor,
                   gamma: torch.Tensor, beta: torch.Tensor, eps: float = 1e-5, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """``LayerNorm(residual + F.linear(x, weight, bias))`` for bf16 [..., 256] tensors and a [256, 256] weight in one kernel
    (MSDA's output_proj + the layer's norm1, ms_deform_attn.py:372-376 / relation_transformer.py:262-271).  Inference only."""
    _require_device(x, weight, bias, residual, gamma, beta, out)
    if not linear_ln_k256_supported(x, weight, residual):
        raise _lib.RdetrError("linear_ln_k256: needs bf16 [..., 256] inputs with evenly strided 16-byte aligned rows and a [256, 256] weight")
    for t in (bias, gamma, beta):
        if t is not None and (t.dtype != torch.bfloat16 or t.numel() != 256):
            raise _lib.RdetrError("linear_ln_k256: bias / gamma / beta must be bf16 [256]")
    rows, _, ldx = _rows_view(x, "linear_ln_k256")
    _, _, ldr = _rows_view(residual, "linear_ln_k256")
    if out is None:
        out = torch.empty(x.shape, dtype=x.dtype, device=x.device)
    elif tuple(out.shape) != tuple(x.shape) or out.dtype != x.dtype:
        raise _lib.RdetrError("linear_ln_k256: out must have x's shape and dtype")
    orows, _, ldo = _rows_view(out, "linear_ln_k256")
    if orows != rows or ldo % 8 or out.data_ptr() % 16:
        raise _lib.RdetrError("linear_ln_k256: out rows must be 16-byte aligned")
    lib = _lib.load()
    def build():
        packed = torch.empty(256 * 256, dtype=torch.bfloat16, device=weight.device)
        _lib.check(lib.rdetr_linear_pack_k256_bf16(weight.data_ptr(), packed.data_ptr(), _stream_ptr(weight)), "rdetr_linear_pack_k256_bf16")
        if not torch.cuda.is_current_stream_capturing():
            torch.cuda.current_stream(weight.device).synchronize()
        return packed
    packed_w = _LINEAR_PACKED.get((weight,), build)
    st = lib.rdetr_linear_ln_k256_bf16(x.data_ptr(), ldx, packed_w.data_ptr(), None if bias is None else _cptr(bias),
                                       residual.data_ptr(), ldr, _cptr(gamma), _cptr(beta),
                                       float(eps), rows, out.data_ptr(), ldo, _stream_ptr(x))
    _lib.check(st, "rdetr_linear_ln_k256_bf16")
    return out
