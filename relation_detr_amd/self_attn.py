"""Decoder self-attention with the position-relation bias (drop-in for ``nn.MultiheadAttention``).

The reference decoder layer (models/bricks/relation_transformer.py:406-408,452-459) calls
``nn.MultiheadAttention(embed_dim, n_heads, dropout, batch_first=True)`` with the float relation bias
``[B*H, N, N]`` as ``attn_mask``.  This module keeps that class's parameter names
(``in_proj_weight, in_proj_bias, out_proj.weight, out_proj.bias`` -- released checkpoints load) and its
call signature for this use, and runs: Wq/Wk/Wv projections and QK^T / PV as dense GEMMs (rocBLAS /
hipBLASLt -> MFMA), and the bias-add + row softmax as one in-place HIP kernel
(``rdetr_bias_softmax_f32``) instead of materialising ``scores + bias`` and the probabilities
separately.
"""
from __future__ import annotations

import math
from typing import Optional

import torch
from torch import Tensor, nn
from torch.nn import functional as F

from . import ops
from .relation import DeferredRelationBias


class _BiasSoftmaxFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, scores, bias, mask):
        probs = ops.bias_softmax_(scores, bias, mask)
        ctx.save_for_backward(probs)
        ctx.has_bias = bias is not None
        ctx.mark_dirty(scores)
        return probs

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, grad):
        (p,) = ctx.saved_tensors
        gs = p * (grad - (grad * p).sum(-1, keepdim=True))
        return gs, (gs if ctx.has_bias else None), None


class RelationSelfAttention(nn.Module):
    """``forward(query, key, value, attn_mask=None, need_weights=False) -> (output, None)`` with
    batch-first ``[B, N, C]`` tensors; ``attn_mask``: float ``[B*H, N, N]`` additive bias (may hold
    -inf), bool ``[N, N]`` (True = masked), None, or a ``DeferredRelationBias`` (the bias as a recipe)."""

    def __init__(self, embed_dim: int, num_heads: int, dropout: float = 0.0, batch_first: bool = True):
        super().__init__()
        if not batch_first:
            raise NotImplementedError("the reference only uses batch_first=True (relation_transformer.py:406-408)")
        if embed_dim % num_heads:
            raise ValueError("embed_dim must be divisible by num_heads")
        self.embed_dim, self.num_heads, self.head_dim = embed_dim, num_heads, embed_dim // num_heads
        self.dropout = dropout
        self.batch_first = True
        self.in_proj_weight = nn.Parameter(torch.empty(3 * embed_dim, embed_dim))
        self.in_proj_bias = nn.Parameter(torch.zeros(3 * embed_dim))
        self.out_proj = nn.Linear(embed_dim, embed_dim)
        nn.init.xavier_uniform_(self.in_proj_weight)
        nn.init.zeros_(self.out_proj.bias)

    def forward(self, query: Tensor, key: Tensor, value: Tensor, attn_mask: Optional[Tensor] = None,
                need_weights: bool = False, key_padding_mask: Optional[Tensor] = None):
        if need_weights or key_padding_mask is not None:
            raise NotImplementedError("need_weights / key_padding_mask are not used on this path")
        B, N, C = query.shape
        M = key.shape[1]
        H, d = self.num_heads, self.head_dim
        if key is query:                                   # the decoder call: q = k = query + pos, one GEMM for both
            qk = F.linear(query, self.in_proj_weight[:2 * C], self.in_proj_bias[:2 * C])
            q, k = qk[..., :C], qk[..., C:]
        else:
            q = F.linear(query, self.in_proj_weight[:C], self.in_proj_bias[:C])
            k = F.linear(key, self.in_proj_weight[C:2 * C], self.in_proj_bias[C:2 * C])
        v = F.linear(value, self.in_proj_weight[2 * C:], self.in_proj_bias[2 * C:])
        needs_grad = torch.is_grad_enabled() and (q.requires_grad or k.requires_grad or v.requires_grad)
        if isinstance(attn_mask, DeferredRelationBias):
            rel = attn_mask
            conv = rel.module.pos_proj[0]
            if (q.is_cuda and q.dtype == torch.bfloat16 and d == 32 and H == 8 and rel.module.num_pos_feats == 16 and not needs_grad
                    and not (torch.is_grad_enabled() and conv.weight.requires_grad)
                    and tuple(rel.src_boxes.shape) == (B, N, 4) and tuple(rel.tgt_boxes.shape) == (B, M, 4)
                    and (rel.attn_mask is None or rel.attn_mask.dtype == torch.bool and tuple(rel.attn_mask.shape) == (N, M))):
                # inference, bf16: the bias is generated inside the attention kernel (csrc/attn_rel.hip), never materialised
                ctx = ops.relation_attention_boxes(q, k, v, H, rel.src_boxes, rel.tgt_boxes, conv.weight, conv.bias, rel.attn_mask,
                                                   1.0 / math.sqrt(d), rel.module.num_pos_feats, rel.module.temperature,
                                                   rel.module.scale)
                return self.out_proj(ctx), None
            attn_mask = rel.materialize()
        if (q.is_cuda and q.dtype == torch.bfloat16 and d == 32 and not needs_grad
                and (attn_mask is None or attn_mask.dtype == torch.bool and attn_mask.dim() == 2
                     or attn_mask.dtype != torch.bool and attn_mask.numel() == B * H * N * M)):
            # inference, bf16: QK^T + bias + softmax + PV in one flash-style kernel (csrc/attn.hip)
            is_bool = attn_mask is not None and attn_mask.dtype == torch.bool
            ctx = ops.relation_attention(q, k, v, H, None if attn_mask is None or is_bool else attn_mask.float(),
                                         attn_mask if is_bool else None, 1.0 / math.sqrt(d))
            return self.out_proj(ctx), None
        # The library's batched GEMMs get DENSE per-head operands [B, H, N, d].  As views of the packed projections (q / k:
        # column slices with row stride 2C, heads interleaved inside a row) the batch of one image folds into a strided-
        # batched GEMM with batch stride d and leading dimension 2C, i.e. matrices that OVERLAP in memory -- legal for the
        # GEMM itself, but PyTorch's TunableOp sizes its scratch copy of such an operand as max(stride * batch, rows * cols
        # * batch) elements (ATen/cuda/tunable/GemmCommon.h, GemmStridedBatchedParams::GetSizeA: the leading dimension is not
        # in the formula), half of what a candidate kernel then reads with lda = 2C: the out-of-bounds read behind round 2's
        # "Memory access fault" while tuning the fp32 decoder.  One copy of N x C elements per operand.
        q = (q * (1.0 / math.sqrt(d))).view(B, N, H, d).transpose(1, 2).contiguous()
        k = k.view(B, M, H, d).transpose(1, 2).contiguous()
        v = v.view(B, M, H, d).transpose(1, 2).contiguous()
        scores = torch.matmul(q, k.transpose(-1, -2)).float().reshape(B * H, N, M).contiguous()
        bias = mask = None
        if attn_mask is not None:
            if attn_mask.dtype == torch.bool:
                if attn_mask.dim() != 2:
                    raise NotImplementedError("boolean attn_mask must be [N, N]")
                mask = attn_mask
            else:
                bias = attn_mask.float().reshape(B * H, N, M)
        probs = _BiasSoftmaxFunction.apply(scores, bias, mask)
        if self.dropout > 0.0 and self.training:
            probs = F.dropout(probs, self.dropout)
        ctx = torch.matmul(probs.view(B, H, N, M).to(v.dtype), v).transpose(1, 2).reshape(B, N, C)
        return self.out_proj(ctx), None
