"""Drop-in ``PositionRelationEmbedding`` / ``box_rel_encoding`` backed by the fused gfx950 kernel.

Contract from the reference (models/bricks/relation_transformer.py:481-532): constructor
``(embed_dim=256, num_heads=8, temperature=10000., scale=100., activation_layer=nn.ReLU, inplace=True)``
(instantiated as ``PositionRelationEmbedding(16, num_heads)``, :301 -- ``embed_dim`` is the number of
sine features per box coordinate), ``forward(src_boxes, tgt_boxes=None) -> [B, H, N1, N2]`` returning a
fresh tensor the caller may mutate (:372-374), state_dict keys ``pos_proj.0.weight [H, 4*embed_dim, 1, 1]``
and ``pos_proj.0.bias [H]``.

The reference materialises [B,N,N,4] -> [B,N,N,64] -> conv -> [B,8,N,N]; here a single kernel
(``rdetr_relation_bias_f32``) goes from the boxes to the bias.  Boxes get no gradient (the
reference wraps the encoding in ``torch.no_grad``, :527-529); ``pos_proj`` does, through a backward
kernel that regenerates the sine features from the boxes (``rdetr_relation_bias_backward_f32``, training only;
other head / feature counts recompute them with device torch ops).
"""
from __future__ import annotations

import torch
from torch import Tensor, nn

from . import ops


def box_rel_encoding(src_boxes: Tensor, tgt_boxes: Tensor, eps: float = 1e-5) -> Tensor:
    """Pairwise log-distance / log-size-ratio encoding [B,N1,N2,4] (relation_transformer.py:481-490).
    Kept for API compatibility and for the training backward; the forward kernel fuses it."""
    c1, s1 = src_boxes[..., None, :2], src_boxes[..., None, 2:] + eps
    c2, s2 = tgt_boxes[..., None, :, :2], tgt_boxes[..., None, :, 2:] + eps
    return torch.cat([torch.log((c1 - c2).abs() / s1 + 1.0), torch.log(s1 / s2)], dim=-1)


def _sine_features(enc: Tensor, num_pos_feats: int, temperature: float, scale: float) -> Tensor:
    """[..., 4] -> [..., 4*F], channel = coord*F + 2k + {sin, cos} (position_encoding.py:115-138, no xy swap)."""
    k = torch.arange(num_pos_feats // 2, dtype=torch.float32, device=enc.device)
    dim_t = temperature ** (k * 2 / num_pos_feats)
    ang = enc.unsqueeze(-1) * scale / dim_t
    return torch.stack((ang.sin(), ang.cos()), dim=-1).flatten(-3)


class _RelationBiasFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, src, tgt, weight, bias, num_pos_feats, temperature, scale):
        out = ops.relation_bias(src, tgt, weight, bias, num_pos_feats, temperature, scale)
        # the caller owns `out` and mutates it in place (`masked_fill_(attn_mask, -inf)`, relation_transformer.py:372-374;
        # the reference returns a clone for the same reason, :530-532): save the ReLU mask, a tensor the caller cannot reach
        ctx.save_for_backward(src, tgt, out > 0)
        ctx.cfg = (num_pos_feats, temperature, scale, weight.shape, bias is not None)
        return out

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, grad_out):
        src, tgt, active = ctx.saved_tensors
        F_, temperature, scale, wshape, has_bias = ctx.cfg
        if ops.relation_bias_backward_supported(wshape[0], F_):
            # one kernel: the features are regenerated from the boxes and reduced on the fly, deterministically
            # (csrc/relation_bwd.hip) -- the reference's autograd keeps [N1, N2, 64] per image for this (207 MB at N = 900)
            gw, gb = ops.relation_bias_backward(src, tgt, grad_out, active, F_, temperature, scale)
            return None, None, gw.view(wshape), (gb if has_bias else None), None, None, None
        g = grad_out * active                                          # ReLU'
        gw = torch.zeros(wshape[0], 4 * F_, dtype=torch.float32, device=g.device)
        for b in range(src.shape[0]):                                  # one image at a time bounds the feature tensor
            feat = _sine_features(box_rel_encoding(src[b:b + 1].float(), tgt[b:b + 1].float()), F_, temperature, scale)
            gw += torch.einsum("hij,ijc->hc", g[b], feat[0])
        gb = g.sum(dim=(0, 2, 3)) if has_bias else None
        return None, None, gw.view(wshape), gb, None, None, None


class DeferredRelationBias:
    """The relation bias of a decoder layer, not materialised: boxes + projection + the optional denoising visibility mask.
    ``RelationSelfAttention`` generates it inside its attention kernel when it can (bf16 inference, 8 heads of 32:
    ``ops.relation_attention_boxes``) and calls ``materialize()`` otherwise.  Made by ``PositionRelationEmbedding.deferred``."""

    def __init__(self, module: "PositionRelationEmbedding", src_boxes: Tensor, tgt_boxes: Tensor, attn_mask: Tensor = None):
        self.module, self.src_boxes, self.tgt_boxes, self.attn_mask = module, src_boxes, tgt_boxes, attn_mask

    def materialize(self) -> Tensor:
        """What the reference hands to the next layer (relation_transformer.py:372-374): [B*H, N1, N2] float, -inf where masked."""
        bias = self.module(self.src_boxes, self.tgt_boxes).flatten(0, 1)
        if self.attn_mask is not None:
            bias.masked_fill_(self.attn_mask, float("-inf"))
        return bias


class PositionRelationEmbedding(nn.Module):
    def __init__(self, embed_dim=256, num_heads=8, temperature=10000.0, scale=100.0, activation_layer=nn.ReLU,
                 inplace=True):
        super().__init__()
        if activation_layer is not nn.ReLU:
            raise NotImplementedError("the fused relation-bias kernel implements the reference's ReLU activation only")
        # index 0 = the 1x1 projection (keeps the reference's state_dict keys pos_proj.0.*), index 1 = activation
        self.pos_proj = nn.Sequential(nn.Conv2d(embed_dim * 4, num_heads, kernel_size=1), nn.ReLU(inplace=inplace))
        self.num_pos_feats = embed_dim
        self.temperature = float(temperature)
        self.scale = float(scale)

    def forward(self, src_boxes: Tensor, tgt_boxes: Tensor = None) -> Tensor:
        if tgt_boxes is None:
            tgt_boxes = src_boxes
        torch._assert(src_boxes.shape[-1] == 4, "src_boxes much have 4 coordinates")
        torch._assert(tgt_boxes.shape[-1] == 4, "tgt_boxes must have 4 coordinates")
        conv = self.pos_proj[0]
        return _RelationBiasFunction.apply(src_boxes.detach(), tgt_boxes.detach(), conv.weight, conv.bias,
                                           self.num_pos_feats, self.temperature, self.scale)

    def deferred(self, src_boxes: Tensor, tgt_boxes: Tensor = None, attn_mask: Tensor = None) -> DeferredRelationBias:
        """Not in the reference: the same bias as a recipe, for an attention kernel that generates it on the fly."""
        tgt_boxes = src_boxes if tgt_boxes is None else tgt_boxes
        torch._assert(src_boxes.shape[-1] == 4, "src_boxes much have 4 coordinates")
        torch._assert(tgt_boxes.shape[-1] == 4, "tgt_boxes must have 4 coordinates")
        return DeferredRelationBias(self, src_boxes.detach(), tgt_boxes.detach(), attn_mask)


PositionRelationEncoder = PositionRelationEmbedding      # the name BASELINE.json's north_star uses
