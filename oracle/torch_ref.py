"""CPU oracle (TEST INFRASTRUCTURE ONLY) for the Relation-DETR hot path.

This file is a plain-PyTorch CPU restatement of the reference's algorithm for the
multi-scale deformable attention (MSDA) + position-relation bias path.  It is the
*checker*: only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline``
leg of ``bench.py`` may import it.  Nothing under ``relation_detr_amd/`` does.

Parity pin: every function here is compared in ``tests/test_oracle_golden.py``
against golden vectors produced by importing the reference's own Python
(``oracle/gen_golden.py``, run in the build container where /root/reference is
mounted) -- see tests/golden/README.md.

The op sequence deliberately uses the same ATen kernels as the reference's
pure-PyTorch fallback (per-level ``grid_sample`` -> stack -> multiply -> sum), so
that timing this file on the GPU box's host cores is a fair stand-in for "the
reference's CPU ms_deform_attn path" (the reference itself cannot travel).

Reference citations (paths relative to /root/reference):
  msda_core            <- models/bricks/ms_deform_attn.py:159-212
  msda_module_forward  <- models/bricks/ms_deform_attn.py:286-377
  box_rel_encoding     <- models/bricks/relation_transformer.py:481-490
  sine_embed           <- models/bricks/position_encoding.py:101-138
  relation_bias        <- models/bricks/relation_transformer.py:493-532
  self_attn_with_bias  <- models/bricks/relation_transformer.py:452-461 (nn.MultiheadAttention)
"""
from __future__ import annotations

import math
from typing import Dict, Optional

import torch
import torch.nn.functional as F


# --------------------------------------------------------------------------- MSDA core
def msda_core(value: torch.Tensor, spatial_shapes, sampling_locations: torch.Tensor,
              attention_weights: torch.Tensor) -> torch.Tensor:
    """value [B,S,H,D]; spatial_shapes [L,2] (h,w); sampling_locations [B,Nq,H,L,P,2] (x,y in 0..1);
    attention_weights [B,Nq,H,L,P]  ->  [B,Nq,H*D]   (ms_deform_attn.py:159-212)."""
    B, S, H, D = value.shape
    Nq, L, P = sampling_locations.shape[1], sampling_locations.shape[3], sampling_locations.shape[4]
    hw = [(int(h), int(w)) for h, w in (spatial_shapes.tolist() if torch.is_tensor(spatial_shapes) else spatial_shapes)]
    per_level = torch.split(value, [h * w for h, w in hw], dim=1)          # level l -> [B, h*w, H, D]
    grid = sampling_locations * 2 - 1                                      # to grid_sample's [-1,1] convention
    sampled = []
    for lvl, (h, w) in enumerate(hw):
        img = per_level[lvl].flatten(2).transpose(1, 2).reshape(B * H, D, h, w)
        g = grid[:, :, :, lvl].transpose(1, 2).flatten(0, 1)               # [B*H, Nq, P, 2]
        sampled.append(F.grid_sample(img, g, mode="bilinear", padding_mode="zeros", align_corners=False))
    weights = attention_weights.transpose(1, 2).reshape(B * H, 1, Nq, L * P)
    stacked = torch.stack(sampled, dim=-2).flatten(-2)                     # [B*H, D, Nq, L*P]
    out = (stacked * weights).sum(-1)                                      # [B*H, D, Nq]
    return out.view(B, H * D, Nq).transpose(1, 2).contiguous()


def sampling_locations_from_reference(reference_points: torch.Tensor, sampling_offsets: torch.Tensor,
                                      spatial_shapes: torch.Tensor, num_points: int) -> torch.Tensor:
    """ms_deform_attn.py:339-349.  reference_points [B,Nq,L,2|4]; sampling_offsets [B,Nq,H,L,P,2]."""
    if reference_points.shape[-1] == 2:
        normalizer = torch.stack([spatial_shapes[..., 1], spatial_shapes[..., 0]], -1)   # (w,h) per level
        return reference_points[:, :, None, :, None, :] + sampling_offsets / normalizer[None, None, None, :, None, :]
    if reference_points.shape[-1] == 4:
        return (reference_points[:, :, None, :, None, :2]
                + sampling_offsets / num_points * reference_points[:, :, None, :, None, 2:] * 0.5)
    raise ValueError("Last dim of reference_points must be 2 or 4, but get {} instead.".format(reference_points.shape[-1]))


def msda_module_forward(params: Dict[str, torch.Tensor], query, reference_points, value, spatial_shapes,
                        level_start_index, key_padding_mask, num_heads=8, num_levels=4, num_points=4):
    """Functional restatement of MultiScaleDeformableAttention.forward (ms_deform_attn.py:286-377).
    ``params`` uses the module's state_dict keys."""
    B, Nq, C = query.shape
    S = value.shape[1]
    v = F.linear(value, params["value_proj.weight"], params["value_proj.bias"])
    if key_padding_mask is not None:
        v = v.masked_fill(key_padding_mask[..., None], 0.0)
    v = v.view(B, S, num_heads, C // num_heads)
    off = F.linear(query, params["sampling_offsets.weight"], params["sampling_offsets.bias"])
    off = off.view(B, Nq, num_heads, num_levels, num_points, 2)
    aw = F.linear(query, params["attention_weights.weight"], params["attention_weights.bias"])
    aw = aw.view(B, Nq, num_heads, num_levels * num_points).softmax(-1)
    aw = aw.view(B, Nq, num_heads, num_levels, num_points)
    loc = sampling_locations_from_reference(reference_points, off, spatial_shapes, num_points)
    core = msda_core(v, spatial_shapes, loc, aw)
    return F.linear(core, params["output_proj.weight"], params["output_proj.bias"])


# --------------------------------------------------------------------------- relation bias
def box_rel_encoding(src_boxes: torch.Tensor, tgt_boxes: torch.Tensor, eps: float = 1e-5) -> torch.Tensor:
    """[B,N1,4],[B,N2,4] cxcywh -> [B,N1,N2,4]   (relation_transformer.py:481-490)."""
    c1, s1 = src_boxes[..., :2], src_boxes[..., 2:]
    c2, s2 = tgt_boxes[..., :2], tgt_boxes[..., 2:]
    dist = (c1[:, :, None, :] - c2[:, None, :, :]).abs()
    dist = torch.log(dist / (s1[:, :, None, :] + eps) + 1.0)
    ratio = torch.log((s1[:, :, None, :] + eps) / (s2[:, None, :, :] + eps))
    return torch.cat([dist, ratio], dim=-1)


def sine_embed(x: torch.Tensor, num_pos_feats: int = 16, temperature: float = 10000.0,
               scale: float = 100.0) -> torch.Tensor:
    """[..., n] -> [..., n*num_pos_feats] with layout coord*F + 2k + {sin:0,cos:1}, no xy exchange
    (position_encoding.py:101-138 as called from relation_transformer.py:512-518)."""
    k = torch.arange(num_pos_feats // 2, dtype=torch.float32, device=x.device)
    dim_t = temperature ** (k * 2 / num_pos_feats)
    a = x.unsqueeze(-1) * scale / dim_t                       # multiply first, then true division
    return torch.stack((a.sin(), a.cos()), dim=-1).flatten(-3)


def relation_bias(src_boxes, tgt_boxes, proj_weight, proj_bias, num_pos_feats: int = 16,
                  temperature: float = 10000.0, scale: float = 100.0) -> torch.Tensor:
    """PositionRelationEmbedding.forward (relation_transformer.py:520-532).
    proj_weight [Hh, 4*num_pos_feats, 1, 1] (or [Hh, 4F]); returns ReLU(conv1x1) as [B,Hh,N1,N2]."""
    if tgt_boxes is None:
        tgt_boxes = src_boxes
    feat = sine_embed(box_rel_encoding(src_boxes, tgt_boxes), num_pos_feats, temperature, scale)
    feat = feat.permute(0, 3, 1, 2)
    w = proj_weight.reshape(proj_weight.shape[0], -1, 1, 1)
    return F.relu(F.conv2d(feat, w, proj_bias)).clone()


# --------------------------------------------------------------------------- decoder self-attention with bias
def bias_softmax(scores: torch.Tensor, bias: Optional[torch.Tensor]) -> torch.Tensor:
    """softmax(scores + bias) over the last dim; scores/bias [B*Hh, N, N]."""
    return torch.softmax(scores if bias is None else scores + bias, dim=-1)


def self_attn_with_bias(q_in, k_in, v_in, in_proj_weight, in_proj_bias, out_proj_weight, out_proj_bias,
                        attn_bias: Optional[torch.Tensor], num_heads: int = 8) -> torch.Tensor:
    """What nn.MultiheadAttention(batch_first=True, dropout=0) computes for the decoder call
    (relation_transformer.py:452-459): softmax(QK^T/sqrt(d) + bias[b*Hh+h]) V, then out_proj.
    q_in/k_in/v_in [B,N,C]; attn_bias float [B*Hh,N,N], bool [N,N] (True = masked) or None."""
    B, N, C = q_in.shape
    d = C // num_heads
    wq, wk, wv = in_proj_weight.chunk(3, dim=0)
    bq, bk, bv = in_proj_bias.chunk(3, dim=0)
    q = F.linear(q_in, wq, bq).view(B, N, num_heads, d).transpose(1, 2)
    k = F.linear(k_in, wk, bk).view(B, -1, num_heads, d).transpose(1, 2)
    v = F.linear(v_in, wv, bv).view(B, -1, num_heads, d).transpose(1, 2)
    scores = (q * (1.0 / math.sqrt(d))) @ k.transpose(-1, -2)               # [B,Hh,N,N]
    if attn_bias is not None:
        if attn_bias.dtype == torch.bool:
            scores = scores.masked_fill(attn_bias, float("-inf"))
        else:
            scores = scores + attn_bias.view(B, num_heads, N, -1)
    ctx = torch.softmax(scores, dim=-1) @ v
    ctx = ctx.transpose(1, 2).reshape(B, N, C)
    return F.linear(ctx, out_proj_weight, out_proj_bias)
