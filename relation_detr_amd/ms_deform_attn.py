"""Drop-in ``MultiScaleDeformableAttention`` backed by the gfx950 HIP kernels.

API contract taken from the reference module (models/bricks/ms_deform_attn.py:215-377): same
constructor ``(embed_dim, num_levels, num_heads, num_points, img2col_step)``, same attributes
(``im2col_step, embed_dim, num_heads, num_levels, num_points``), same sub-module names -- hence the
same state_dict keys ``sampling_offsets / attention_weights / value_proj / output_proj`` (the
optimiser grouping matches on ``sampling_offsets``, optimizer/param_dict.py:82) -- same
``init_weights()`` and the same keyword ``forward``.  The four projections stay ``nn.Linear``
(rocBLAS / hipBLASLt, i.e. MFMA); the gather-and-weighted-sum core is ``rdetr_msda_forward_*``.
"""
from __future__ import annotations

import math
import warnings

import torch
from torch import Tensor, nn

from . import ops
from . import options as _options


def sampling_locations(reference_points: Tensor, offsets: Tensor, spatial_shapes: Tensor, num_points: int) -> Tensor:
    """reference_points [B,Nq,L,2|4], offsets [B,Nq,H,L,P,2] -> normalised (x, y) locations
    [B,Nq,H,L,P,2]  (ms_deform_attn.py:339-349)."""
    last = reference_points.shape[-1]
    if last == 2:
        wh = spatial_shapes.flip(-1).to(offsets.dtype)                          # (w, h) per level
        return reference_points[:, :, None, :, None, :] + offsets / wh[None, None, None, :, None, :]
    if last == 4:
        centre = reference_points[:, :, None, :, None, :2]
        size = reference_points[:, :, None, :, None, 2:]
        return centre + offsets / num_points * size * 0.5
    raise ValueError("Last dim of reference_points must be 2 or 4, but get {} instead.".format(last))


class MultiScaleDeformableAttention(nn.Module):
    def __init__(self, embed_dim: int = 256, num_levels: int = 4, num_heads: int = 8, num_points: int = 4,
                 img2col_step: int = 64):
        super().__init__()
        if embed_dim % num_heads != 0:
            raise ValueError(
                "embed_dim must be divisible by num_heads, but got {} and {}".format(embed_dim, num_heads))
        head_dim = embed_dim // num_heads
        if head_dim & (head_dim - 1):
            warnings.warn("head_dim is not a power of two; the wave-per-query HIP kernel needs head_dim == 32, "
                          "other sizes run the generic kernel")
        self.im2col_step = img2col_step            # accepted for compatibility; the HIP op has no batch chunking
        self.embed_dim = embed_dim
        self.num_heads = num_heads
        self.num_levels = num_levels
        self.num_points = num_points
        self.options = _options.get()          # kernel-routing switches, fixed at construction (options.py)
        self.sampling_offsets = nn.Linear(embed_dim, num_heads * num_levels * num_points * 2)
        self.attention_weights = nn.Linear(embed_dim, num_heads * num_levels * num_points)
        self.value_proj = nn.Linear(embed_dim, embed_dim)
        self.output_proj = nn.Linear(embed_dim, embed_dim)
        self.init_weights()

    def init_weights(self):
        """Reference initialisation (ms_deform_attn.py:266-284): zero offset weights with a ring of
        unit directions per head scaled by the point index as bias, uniform attention, xavier projections."""
        nn.init.zeros_(self.sampling_offsets.weight)
        angle = torch.arange(self.num_heads, dtype=torch.float32) * (2.0 * math.pi / self.num_heads)
        ring = torch.stack([angle.cos(), angle.sin()], -1)
        ring = ring / ring.abs().max(-1, keepdim=True)[0]
        ring = ring.view(self.num_heads, 1, 1, 2).repeat(1, self.num_levels, self.num_points, 1)
        ring = ring * torch.arange(1, self.num_points + 1, dtype=torch.float32).view(1, 1, self.num_points, 1)
        with torch.no_grad():
            self.sampling_offsets.bias = nn.Parameter(ring.reshape(-1))
        nn.init.zeros_(self.attention_weights.weight)
        nn.init.zeros_(self.attention_weights.bias)
        nn.init.xavier_uniform_(self.value_proj.weight)
        nn.init.zeros_(self.value_proj.bias)
        nn.init.xavier_uniform_(self.output_proj.weight)
        nn.init.zeros_(self.output_proj.bias)

    def _merged_query_projection(self):
        """sampling_offsets and attention_weights as ONE [3*H*L*P, C] projection (inference): both read the same query, so
        one GEMM writes [rows, 2*H*L*P | H*L*P] and the kernel reads the two column slices in place.  Cached until a
        parameter changes (the separate nn.Linear modules stay the owners: state_dict keys are the reference's)."""
        so, aw = self.sampling_offsets, self.attention_weights
        key = (so.weight._version, so.bias._version, aw.weight._version, aw.bias._version, so.weight.data_ptr(),
               aw.weight.data_ptr(), so.weight.dtype, so.weight.device)
        cache = getattr(self, "_merged_cache", None)
        if cache is None or cache[0] != key:
            with torch.no_grad():
                cache = (key, torch.cat([so.weight, aw.weight], 0).contiguous(), torch.cat([so.bias, aw.bias], 0).contiguous())
            if so.weight.is_cuda and not torch.cuda.is_current_stream_capturing():
                torch.cuda.current_stream(so.weight.device).synchronize()      # other streams (image groups) use it without an event
            object.__setattr__(self, "_merged_cache", cache)
        return cache[1], cache[2]

    def _projections(self, query: Tensor, value: Tensor, key_padding_mask, fill: bool = True, merged: bool = False,
                     want_value: bool = True):
        """value projection (+ padding zero-fill, ms_deform_attn.py:316-321, unless the caller applies the mask itself)
        and the two raw query projections."""
        B, Nq, _ = query.shape
        S = value.shape[1]
        H, L, P = self.num_heads, self.num_levels, self.num_points
        if not want_value:
            v = None
        elif value.dim() == 3 and not value.is_contiguous() and value.stride(2) == 1 and value.stride(0) == S * value.stride(1):
            # a column slice of a wider row-major buffer: as a 2-d strided matrix the projection stays ONE GEMM with the bias
            # in its epilogue (on a non-contiguous 3-d input nn.Linear runs matmul + a separate bias pass)
            v = torch.nn.functional.linear(value.view(B * S, value.shape[2]), self.value_proj.weight,
                                           self.value_proj.bias).view(B, S, -1)
        else:
            v = self.value_proj(value)
        if v is not None and key_padding_mask is not None and fill:
            if torch.is_grad_enabled() and v.requires_grad:
                v = v.masked_fill(key_padding_mask[..., None], float(0))
            elif v.is_cuda and v.dtype in (torch.float32, torch.bfloat16) and v.is_contiguous():
                ops.zero_masked_rows_(v, key_padding_mask)          # inference: write only the padded rows of the fresh projection
            else:
                v.masked_fill_(key_padding_mask[..., None], float(0))
        if v is not None:
            v = v.view(B, S, H, self.embed_dim // H)
        if merged:
            w, b = self._merged_query_projection()
            both = torch.nn.functional.linear(query, w, b)                      # [B, Nq, 3*H*L*P]
            n_off = H * L * P * 2
            offsets = both[..., :n_off].view(B, Nq, H, L, P, 2)                 # column slices, read in place by the kernel
            logits = both[..., n_off:].view(B, Nq, H, L * P)
            return v, offsets, logits
        offsets = self.sampling_offsets(query).view(B, Nq, H, L, P, 2)
        logits = self.attention_weights(query).view(B, Nq, H, L * P)
        return v, offsets, logits

    def project_inputs(self, query: Tensor, reference_points: Tensor, value: Tensor, spatial_shapes: Tensor,
                       key_padding_mask):
        """Everything before the core, materialised (the reference's own sequence, ms_deform_attn.py:316-349):
        projected value, sampling locations, soft-maxed weights.  Pure torch, device-agnostic."""
        v, offsets, logits = self._projections(query, value, key_padding_mask)
        weights = logits.softmax(-1).view(*offsets.shape[:5])
        return v, sampling_locations(reference_points, offsets, spatial_shapes, self.num_points), weights

    def forward(self, query: Tensor, reference_points: Tensor, value: Tensor, spatial_shapes: Tensor,
                level_start_index: Tensor, key_padding_mask: Tensor, post_norm=None, projected_value: Tensor = None) -> Tensor:
        """query [B,Nq,C]; reference_points [B,Nq,L,2] or [B,Nq,L,4]; value [B,S,C]; spatial_shapes
        [L,2] (h,w); level_start_index [L]; key_padding_mask [B,S] bool or None -> [B,Nq,C].
        ``post_norm = (residual, layer_norm)`` (not in the reference's signature, optional): return
        ``layer_norm(residual + output)`` -- the caller's next two steps (relation_transformer.py:270-271) -- which tall bf16
        inference inputs get from the output projection's own epilogue (csrc/linear.hip).
        ``projected_value`` (not in the reference's signature, optional; bf16 inference): ``value_proj(value)`` WITHOUT the padding
        fill, already computed by the caller -- a [B,S,C] view that may be a column slice of a wider buffer (the decoder runs the
        value projections of its six layers as one GEMM, transformer.py).  Ignored unless the fused kernel path applies."""
        if value.is_cuda:      # same consistency check as the reference (:313), from a cached host copy
            shapes, _ = ops.host_levels(spatial_shapes, level_start_index)
            assert sum(h * w for h, w in shapes) == value.shape[1]
        if reference_points.shape[-1] not in (2, 4):
            raise ValueError(
                "Last dim of reference_points must be 2 or 4, but get {} instead.".format(reference_points.shape[-1]))
        fused = (value.is_cuda and not torch.is_grad_enabled()
                 and ops.msda_fast_path(self.num_heads, self.embed_dim // self.num_heads, self.num_levels, self.num_points))
        # the padding mask inside the kernel costs 4 byte loads per sample: cheaper than a fill pass over the projected
        # value for the decoder's few hundred queries, dearer for the encoder's Nq == S (measured: +32 us vs -21 us per call)
        mask_in_kernel = fused and key_padding_mask is not None and query.shape[1] * 4 <= value.shape[1]
        _force = self.options.mask_in_kernel                     # A/B aid: "always" / "never"
        if _force and fused and key_padding_mask is not None:
            mask_in_kernel = _force == "always"
        # encoder shape (queries = the pyramid's own pixels) in bf16: the gather runs on a head-major value [B,H,S,D] (contiguous
        # head planes: 137 -> 115 us at BASELINE.json configs[1]), any level count; the padding zero-fill is
        # folded into the producer of that layout (no fill pass of its own)
        head_major = (fused and value.dtype == torch.bfloat16 and query.shape[1] == value.shape[1]
                      and value.shape[1] >= 4096 and self.options.value_head_major)
        # ... written by the value projection itself where the hand-written projection kernel applies (csrc/linear.hip)
        proj_hm = (head_major and self.options.value_proj_hm and self.value_proj.bias is not None
                   and self.value_proj.bias.dtype == torch.bfloat16 and ops.linear_k256_supported(value, self.value_proj.weight))
        pre = (projected_value is not None and fused and not head_major and value.dtype == torch.bfloat16
               and (key_padding_mask is None or mask_in_kernel) and projected_value.dtype == torch.bfloat16
               and tuple(projected_value.shape) == tuple(value.shape))
        fused_in = None
        if proj_hm and self.options.encoder_proj and self.options.merged_proj and self.num_levels in (4, 5) and query.shape == value.shape:
            # encoder layer, bf16: value_proj (head-major, padded rows zero) and the merged offsets | logits projection of
            # `query` in ONE kernel (csrc/proj.hip) instead of two launches of ~20 us
            wq, bq = self._merged_query_projection()
            if ops.encoder_proj_supported(value, query, self.value_proj.weight, wq):
                fused_in = ops.encoder_proj(value, query, self.value_proj.weight, self.value_proj.bias, wq, bq, key_padding_mask)
        if fused_in is not None:
            B_, Nq_ = query.shape[:2]
            H_, L_, P_ = self.num_heads, self.num_levels, self.num_points
            n_off = H_ * L_ * P_ * 2
            v, offsets, logits = None, fused_in[1][..., :n_off].view(B_, Nq_, H_, L_, P_, 2), fused_in[1][..., n_off:].view(B_, Nq_, H_, L_ * P_)
        else:
            v, offsets, logits = self._projections(query, value, key_padding_mask, fill=not (mask_in_kernel or head_major),
                                                  merged=fused and self.options.merged_proj,
                                                  want_value=not (proj_hm or pre))
        if pre:
            v = projected_value.view(*value.shape[:2], self.num_heads, self.embed_dim // self.num_heads)
        vdt = value.dtype if proj_hm else v.dtype
        core_dtype = vdt if vdt in (torch.float32, torch.bfloat16) else torch.float32
        needs_grad = torch.is_grad_enabled() and any(
            t is not None and t.requires_grad for t in (v, offsets, logits, reference_points))
        if fused and head_major and vdt == torch.bfloat16:
            if fused_in is not None:
                vh = fused_in[0]
            elif proj_hm:
                vh = ops.value_proj_head_major(value, self.value_proj.weight, self.value_proj.bias, key_padding_mask)
            else:
                vh = ops.value_to_head_major(v.view(v.shape[0], v.shape[1], -1), key_padding_mask)
            core = ops.ms_deform_attn_forward_fused(vh, spatial_shapes, level_start_index, offsets, logits,
                                                    reference_points.float().contiguous(), None, value_layout="bhsd")
        elif fused:
            # inference: softmax + location arithmetic happen inside the gather kernel's set-up phase, and so does the
            # padding mask (rows of padded positions count as zero: no fill pass over the projected value)
            core = ops.ms_deform_attn_forward_fused(
                v if pre else v.to(core_dtype).contiguous(), spatial_shapes, level_start_index, offsets.to(core_dtype),
                logits.to(core_dtype), reference_points.float().contiguous(),
                key_padding_mask if mask_in_kernel else None)
        elif (v.is_cuda and not needs_grad
                and ops.msda_fast_path(self.num_heads, self.embed_dim // self.num_heads, self.num_levels, self.num_points)):
            core = ops.ms_deform_attn_forward_fused(
                v.to(core_dtype).contiguous(), spatial_shapes, level_start_index, offsets.to(core_dtype).contiguous(),
                logits.to(core_dtype).contiguous(), reference_points.float().contiguous())
        else:
            weights = logits.softmax(-1).view(*offsets.shape[:5])
            loc = sampling_locations(reference_points, offsets, spatial_shapes, self.num_points)
            core = ops.MultiScaleDeformableAttnFunction.apply(
                v.to(core_dtype).contiguous(), spatial_shapes, level_start_index, loc.float().contiguous(),
                weights.float().contiguous(), self.im2col_step)
        if core.dtype != vdt:
            core = core.to(vdt)
        if post_norm is None:
            return self.output_proj(core)
        residual, norm = post_norm
        if (core.is_cuda and not torch.is_grad_enabled() and core.numel() // core.shape[-1] >= 16384
                and self.options.proj_ln and norm.weight is not None and norm.bias is not None
                and ops.linear_ln_k256_supported(core, self.output_proj.weight, residual)):
            return ops.linear_ln_k256(core, self.output_proj.weight, self.output_proj.bias, residual, norm.weight, norm.bias, norm.eps)
        out = self.output_proj(core)
        if out.is_cuda and not torch.is_grad_enabled() and out.dtype in (torch.float32, torch.bfloat16):
            return ops.add_layer_norm(out, residual, norm.weight, norm.bias, norm.eps)
        return norm(residual + out)
