"""Transformer-level harness around the hot-path modules -- INFERENCE path.

The callers either side of the hot path, restated so that an image batch can be run end to end from
multi-level feature pyramids to per-layer class logits / boxes:
    encoder layer / encoder      models/bricks/relation_transformer.py:162-276
    decoder layer / decoder      models/bricks/relation_transformer.py:279-478
    two-stage transformer        models/bricks/relation_transformer.py:16-159, models/bricks/base_transformer.py
Parameter names follow the reference (`encoder.layers.N.self_attn.*`, `decoder.layers.N.cross_attn.*`,
`decoder.position_relation_embedding.pos_proj.0.*`, `level_embeds`, `enc_output`, ...), so a reference
`RelationTransformer.state_dict()` loads with `load_state_dict`.  What is NOT here: the training-only branches
(denoising queries, the hybrid one-to-many decoder pass) -- `forward` is the eval path
(relation_transformer.py:59-159 with `self.training == False` and no noised queries).

The three hot-path module classes are injectable (`msda_cls`, `self_attn_cls`, `relation_cls`) so that bench.py can
time the same glue with the CPU oracle's operators as the host baseline; the defaults are the HIP-backed modules.
"""
from __future__ import annotations

import math
from typing import List, Sequence

import torch
from torch import Tensor, nn
from torch.nn import functional as F

from . import ops
from . import options as _options
from .ms_deform_attn import MultiScaleDeformableAttention
from .relation import PositionRelationEmbedding
from .self_attn import RelationSelfAttention


def inverse_sigmoid(x: Tensor, eps: float = 1e-3) -> Tensor:
    """util/misc.py:31-35."""
    x = x.clamp(min=0, max=1)
    return torch.log(x.clamp(min=eps) / (1 - x).clamp(min=eps))


def refine_boxes(delta: Tensor, reference: Tensor) -> Tensor:
    """sigmoid(delta + inverse_sigmoid(reference)) -- the iterative box refinement (relation_transformer.py:363-381); boxes stay
    fp32 whatever the network dtype.  One HIP kernel on a device without autograd, torch otherwise."""
    if delta.is_cuda and reference.dtype == torch.float32 and not (torch.is_grad_enabled() and delta.requires_grad):
        from . import ops
        return ops.box_refine(delta, reference)
    return (delta.float() + inverse_sigmoid(reference)).sigmoid()


def sine_pos_embed(pos: Tensor, num_pos_feats: int = 128, temperature: float = 10000.0,
                   scale: float = 2 * math.pi) -> Tensor:
    """get_sine_pos_embed with exchange_xy=True (models/bricks/position_encoding.py:115-138):
    [..., n] -> [..., n*num_pos_feats], first two coordinates swapped (y before x)."""
    k = torch.arange(num_pos_feats // 2, dtype=torch.float32, device=pos.device)
    dim_t = temperature ** (k * 2 / num_pos_feats)
    ang = pos.unsqueeze(-1) * scale / dim_t
    emb = torch.stack((ang.sin(), ang.cos()), dim=-1).flatten(-2)            # [..., n, F]
    # swap the first two coordinates with slices (an index list would be uploaded from the host on every call,
    # which a HIP-graph capture does not allow)
    return torch.cat((emb[..., 1:2, :], emb[..., 0:1, :], emb[..., 2:, :]), dim=-2).flatten(-2)


def add_norm(norm: nn.LayerNorm, x: Tensor, residual: Tensor = None, out: Tensor = None) -> Tensor:
    """norm(x + residual).  On a ROCm device without autograd: one HIP kernel (rdetr_add_layernorm_*, csrc/layernorm.hip)
    instead of an add pass and a normalisation pass; otherwise (CPU oracle harness, training) plain torch."""
    needs_grad = torch.is_grad_enabled() and (x.requires_grad or norm.weight.requires_grad or
                                              (residual is not None and residual.requires_grad))
    if x.is_cuda and not needs_grad and x.dtype in (torch.float32, torch.bfloat16):
        from . import ops
        return ops.add_layer_norm(x, residual, norm.weight, norm.bias, norm.eps, out=out)
    y = norm(x if residual is None else x + residual)
    if out is not None:
        out.copy_(y)
        return out
    return y


_K256_MIN_ROWS = 16384          # below this the library GEMM's shorter fixed cost wins (csrc/linear.hip)


def linear_relu(linear: nn.Linear, x: Tensor, opts: "_options.Options" = None) -> Tensor:
    """relu(linear(x)); on a device the ReLU runs in the GEMM's epilogue (hipBLASLt) instead of a pass of its own over
    the [.., d_ffn] activations."""
    if x.is_cuda and linear.bias is not None and not (torch.is_grad_enabled() and (x.requires_grad or linear.weight.requires_grad)):
        rows = x.numel() // x.shape[-1]
        if (rows >= _K256_MIN_ROWS and linear.out_features >= 1024 and (opts or _options.get()).linear_k256
                and not torch.is_grad_enabled() and ops.linear_k256_supported(x, linear.weight)):
            # opt-in: hand-written MFMA kernel for the tall K = 256, wide-output linear1 of the encoder FFN.  Alone it beats the
            # library GEMM (72 vs 87 us at 44,646 rows), inside the two-group replay it does not (-1 %: its persistent
            # 128-KiB-LDS workgroups leave the other group's kernels no room), hence not the default
            return ops.linear_k256(x, linear.weight, linear.bias, relu=True)
        y = torch._addmm_activation(linear.bias, x.reshape(-1, x.shape[-1]), linear.weight.t(), use_gelu=False)
        return y.view(*x.shape[:-1], linear.out_features)
    return F.relu(linear(x))


def _fused_ffn_applies(linear1: nn.Linear, linear2: nn.Linear, x: Tensor, opts: "_options.Options" = None) -> bool:
    return (x.is_cuda and not torch.is_grad_enabled() and x.numel() // x.shape[-1] >= _K256_MIN_ROWS
            and linear1.bias is not None and linear2.bias is not None and (opts or _options.get()).ffn_fused
            and ops.ffn_k256_supported(x, linear1.weight, linear2.weight))


def feed_forward(linear1: nn.Linear, linear2: nn.Linear, x: Tensor, opts: "_options.Options" = None) -> Tensor:
    """linear2(relu(linear1(x))) (relation_transformer.py:226-233, 272-275).  Tall bf16 inputs at inference go through the fused
    kernel (csrc/ffn.hip: the [rows, d_ffn] activations never reach HBM); options.ffn_fused = False keeps the two library GEMMs."""
    if _fused_ffn_applies(linear1, linear2, x, opts):
        return ops.ffn_k256(x, linear1.weight, linear1.bias, linear2.weight, linear2.bias)
    return linear2(linear_relu(linear1, x, opts))


class MLP(nn.Module):
    """models/bricks/basic.py:6-24 (ReLU between layers, none after the last)."""

    def __init__(self, input_dim: int, hidden_dim: int, output_dim: int, num_layers: int):
        super().__init__()
        dims = [input_dim] + [hidden_dim] * (num_layers - 1) + [output_dim]
        self.num_layers = num_layers
        self.options = _options.get()
        self.layers = nn.ModuleList(nn.Linear(a, b) for a, b in zip(dims[:-1], dims[1:]))
        for layer in self.layers:
            nn.init.xavier_uniform_(layer.weight)
            nn.init.zeros_(layer.bias)

    def forward(self, x: Tensor) -> Tensor:
        for i, layer in enumerate(self.layers):
            x = linear_relu(layer, x, self.options) if i + 1 < self.num_layers else layer(x)
        return x


class RelationTransformerEncoderLayer(nn.Module):
    def __init__(self, embed_dim=256, d_ffn=1024, n_heads=8, n_levels=4, n_points=4, msda_cls=MultiScaleDeformableAttention):
        super().__init__()
        self.embed_dim = embed_dim
        self.options = _options.get()
        self.self_attn = msda_cls(embed_dim, n_levels, n_heads, n_points)
        self.norm1 = nn.LayerNorm(embed_dim)
        self.linear1 = nn.Linear(embed_dim, d_ffn)
        self.linear2 = nn.Linear(d_ffn, embed_dim)
        self.norm2 = nn.LayerNorm(embed_dim)
        nn.init.xavier_uniform_(self.linear1.weight)
        nn.init.xavier_uniform_(self.linear2.weight)

    def forward(self, query, query_pos, reference_points, spatial_shapes, level_start_index, key_padding_mask=None, out=None,
                query_plus_pos=None, next_pos=None):
        """`out`: optional destination of the layer's output (a column slice of the encoder's memory-fusion input).
        `query_plus_pos`: `query + query_pos` if the caller already has it; `next_pos`: also return output + next_pos
        (the next layer's `query_plus_pos`), produced by the final add+LayerNorm pass -> (output, output + next_pos)."""
        if query_plus_pos is None:
            query_plus_pos = query if query_pos is None else query + query_pos
        if type(self.self_attn).forward is MultiScaleDeformableAttention.forward:       # not a subclass with its own forward
            query = self.self_attn(query=query_plus_pos, reference_points=reference_points, value=query,
                                   spatial_shapes=spatial_shapes, level_start_index=level_start_index,
                                   key_padding_mask=key_padding_mask, post_norm=(query, self.norm1))     # norm1(query + attn)
        else:
            attn = self.self_attn(query=query_plus_pos, reference_points=reference_points,
                                  value=query, spatial_shapes=spatial_shapes, level_start_index=level_start_index,
                                  key_padding_mask=key_padding_mask)
            query = add_norm(self.norm1, query, attn)
        if self.options.ffn_ln and _fused_ffn_applies(self.linear1, self.linear2, query, self.options):
            # opt-in: feed-forward block, residual, LayerNorm (and the next layer's query + pos) in ONE kernel (csrc/ffn.hip).
            # Correct (tests/test_gpu_glue.py) but 3-6 % slower in the stack than fused FFN + the add+LayerNorm kernel: the
            # epilogue runs with the matrix pipe idle, the separate kernel overlaps the other image group
            return ops.ffn_ln_k256(query, self.linear1.weight, self.linear1.bias, self.linear2.weight, self.linear2.bias,
                                   self.norm2.weight, self.norm2.bias, self.norm2.eps, out=out, pos=next_pos)
        ffn = feed_forward(self.linear1, self.linear2, query, self.options)
        if next_pos is not None:
            return ops.add_layer_norm(query, ffn, self.norm2.weight, self.norm2.bias, self.norm2.eps, out=out, pos=next_pos)
        return add_norm(self.norm2, query, ffn, out=out)


class RelationTransformerEncoder(nn.Module):
    """6 layers, then fuse the input and every layer's output: cat -> Linear -> ReLU -> Linear -> LayerNorm."""

    def __init__(self, layers: Sequence[RelationTransformerEncoderLayer]):
        super().__init__()
        self.layers = nn.ModuleList(layers)
        self.num_layers = len(self.layers)
        self.embed_dim = self.layers[0].embed_dim
        self.options = _options.get()
        d = self.embed_dim
        self.memory_fusion = nn.Sequential(nn.Linear((self.num_layers + 1) * d, d), nn.ReLU(inplace=True), nn.Linear(d, d),
                                           nn.LayerNorm(d))

    def forward(self, query, spatial_shapes, level_start_index, reference_points, query_pos=None, query_key_padding_mask=None,
                fusion_buffer=None):
        """`fusion_buffer`: optional [B, S, 7*d] buffer whose first column block IS `query` (the caller wrote the tokens there)."""
        fuse = self.memory_fusion
        if query.is_cuda and not torch.is_grad_enabled():
            # inference: every layer writes its output straight into its column slice of the fusion input (the
            # reference concatenates the 7 tensors afterwards, relation_transformer.py:212); the slices are then
            # the next layer's input as strided views
            d = self.embed_dim
            if fusion_buffer is not None:
                stacked = fusion_buffer
            else:
                stacked = torch.empty(*query.shape[:-1], (self.num_layers + 1) * d, dtype=query.dtype, device=query.device)
                stacked[..., :d].copy_(query)
            query = stacked[..., :d]
            fuse_pos = (query_pos is not None and query.dtype in (torch.float32, torch.bfloat16)
                        and self.options.ln_pos)                                # False: separate add (A/B)
            qpp = None
            for i, layer in enumerate(self.layers):
                last = i + 1 == self.num_layers
                res = layer(query, query_pos, reference_points, spatial_shapes, level_start_index, query_key_padding_mask,
                            out=stacked[..., (i + 1) * d:(i + 2) * d], query_plus_pos=qpp,
                            next_pos=query_pos if fuse_pos and not last else None)
                query, qpp = res if isinstance(res, tuple) else (res, None)
            return add_norm(fuse[3], fuse[2](linear_relu(fuse[0], stacked, self.options)))
        outs = [query]
        for layer in self.layers:
            query = layer(query, query_pos, reference_points, spatial_shapes, level_start_index, query_key_padding_mask)
            outs.append(query)
        return add_norm(fuse[3], fuse[2](linear_relu(fuse[0], torch.cat(outs, -1), self.options)))


class RelationTransformerDecoderLayer(nn.Module):
    def __init__(self, embed_dim=256, d_ffn=1024, n_heads=8, n_levels=4, n_points=4,
                 msda_cls=MultiScaleDeformableAttention, self_attn_cls=RelationSelfAttention):
        super().__init__()
        self.embed_dim, self.num_heads = embed_dim, n_heads
        self.options = _options.get()
        self.cross_attn = msda_cls(embed_dim, n_levels, n_heads, n_points)
        self.norm1 = nn.LayerNorm(embed_dim)
        self.self_attn = self_attn_cls(embed_dim, n_heads, dropout=0.0, batch_first=True)
        self.norm2 = nn.LayerNorm(embed_dim)
        self.linear1 = nn.Linear(embed_dim, d_ffn)
        self.linear2 = nn.Linear(d_ffn, embed_dim)
        self.norm3 = nn.LayerNorm(embed_dim)
        nn.init.xavier_uniform_(self.linear1.weight)
        nn.init.xavier_uniform_(self.linear2.weight)

    def forward(self, query, query_pos, reference_points, value, spatial_shapes, level_start_index, self_attn_mask=None,
                key_padding_mask=None, query_plus_pos=None, projected_value=None):
        """``query_plus_pos`` (not in the reference's signature, optional): ``query + query_pos`` if the caller already has it;
        ``projected_value``: ``cross_attn.value_proj(value)`` if the caller already has it (see the decoder)."""
        qp = query + query_pos if query_plus_pos is None else query_plus_pos
        attn = self.self_attn(query=qp, key=qp, value=query, attn_mask=self_attn_mask, need_weights=False)[0]
        if (query.is_cuda and not torch.is_grad_enabled() and query.dtype in (torch.float32, torch.bfloat16)
                and query_pos.dtype == query.dtype and query_pos.shape == query.shape and self.options.decoder_ln_pos):
            # inference: norm2 and the cross-attention's `query + query_pos` from one pass (csrc/layernorm.hip)
            query, cross_q = ops.add_layer_norm(attn, query, self.norm2.weight, self.norm2.bias, self.norm2.eps, pos=query_pos)
        else:
            query = add_norm(self.norm2, query, attn)
            cross_q = query + query_pos
        cross = self.cross_attn(query=cross_q, reference_points=reference_points, value=value,
                                spatial_shapes=spatial_shapes, level_start_index=level_start_index,
                                key_padding_mask=key_padding_mask,
                                **({} if projected_value is None else {"projected_value": projected_value}))
        query = add_norm(self.norm1, query, cross)
        return add_norm(self.norm3, query, feed_forward(self.linear1, self.linear2, query, self.options))


class RelationTransformerDecoder(nn.Module):
    """Iterative box refinement with the position-relation bias between layers (relation_transformer.py:320-383)."""

    def __init__(self, layers: Sequence[RelationTransformerDecoderLayer], num_classes: int,
                 relation_cls=PositionRelationEmbedding):
        super().__init__()
        self.layers = nn.ModuleList(layers)
        self.num_layers = len(self.layers)
        self.embed_dim, self.num_heads = self.layers[0].embed_dim, self.layers[0].num_heads
        self.options = _options.get()
        d = self.embed_dim
        self.ref_point_head = MLP(2 * d, d, d, 2)
        self.query_scale = MLP(d, d, d, 2)
        self.class_head = nn.ModuleList(nn.Linear(d, num_classes) for _ in range(self.num_layers))
        self.bbox_head = nn.ModuleList(MLP(d, d, 4, 3) for _ in range(self.num_layers))
        self.norm = nn.LayerNorm(d)
        self.position_relation_embedding = relation_cls(16, self.num_heads)
        prior = -math.log((1 - 0.01) / 0.01)
        for head in self.class_head:
            nn.init.constant_(head.bias, prior)
        for head in self.bbox_head:
            nn.init.zeros_(head.layers[-1].weight)
            nn.init.zeros_(head.layers[-1].bias)

    def _batched_value_projection(self, value: Tensor):
        """``cross_attn.value_proj(value)`` of ALL layers as one GEMM: [B,S,C] x [C, layers*C] -> [B,S,layers*C]; layer l reads its
        column slice in place (csrc/msda_fwd.hip takes the pixel stride).  The reference runs the six projections one per layer
        on the same encoder memory (relation_transformer.py:464-471 -> ms_deform_attn.py:316); batched they are one chip-filling
        launch ahead of the decoder's chain of small launches instead of six launches on it.  OPT-IN (options.decoder_value_batched):
        measured 1 % SLOWER in the two-group replay (978 vs 988 images/s, same box) -- the batched GEMM stays on the decoder's
        own stream, so it saves at most the start-up of five launches, and one long chip-filling launch in front of the chain
        is worse for the other image group than six short ones along it; it would pay where the projections can run BESIDE
        the chain.  The weights stay owned by the
        layers' nn.Linear modules (state_dict keys unchanged); the concatenation is cached until one of them changes."""
        projs = [layer.cross_attn.value_proj for layer in self.layers]
        key = tuple((p.weight._version, p.bias._version, p.weight.data_ptr(), p.weight.dtype) for p in projs)
        cache = getattr(self, "_value_proj_cache", None)
        if cache is None or cache[0] != key:
            with torch.no_grad():
                cache = (key, torch.cat([p.weight for p in projs], 0).contiguous(), torch.cat([p.bias for p in projs], 0).contiguous())
            if value.is_cuda and not torch.cuda.is_current_stream_capturing():
                torch.cuda.current_stream(value.device).synchronize()         # other streams (image groups) use it without an event
            object.__setattr__(self, "_value_proj_cache", cache)
        B, S, C = value.shape
        v2 = value if value.is_contiguous() else value.contiguous()
        return F.linear(v2.view(B * S, C), cache[1], cache[2]).view(B, S, len(projs) * C)

    def forward(self, query, reference_points, value, spatial_shapes, level_start_index, valid_ratios,
                key_padding_mask=None, attn_mask=None, skip_relation=False):
        classes: List[Tensor] = []
        coords: List[Tensor] = []
        ratio_scale = None                                                          # [B,1,L,4], built when the torch path needs it
        pos_relation = attn_mask
        tgt_boxes = None
        values_all = None
        if (self.options.decoder_value_batched and value.is_cuda and not torch.is_grad_enabled() and value.dtype == torch.bfloat16
                and query.shape[1] * 4 <= value.shape[1] and value.shape[-1] == self.embed_dim
                and all(type(l.cross_attn).forward is MultiScaleDeformableAttention.forward and l.cross_attn.value_proj.bias is not None
                        for l in self.layers)
                and ops.msda_fast_path(self.num_heads, self.embed_dim // self.num_heads, spatial_shapes.shape[0], 4)):
            values_all = self._batched_value_projection(value)                  # [B, S, layers * C]
        for idx, layer in enumerate(self.layers):
            if (reference_points.is_cuda and not torch.is_grad_enabled() and query.dtype in (torch.float32, torch.bfloat16)
                    and reference_points.dtype == torch.float32 and valid_ratios.dtype == torch.float32
                    and self.options.decoder_entry):
                # inference: the scaling by the valid ratios, the level-0 slice and its sine embedding in one launch (csrc/glue.hip)
                ref_in, emb = ops.decoder_reference(reference_points.detach(), valid_ratios, self.embed_dim // 2, dtype=query.dtype)
            else:
                if ratio_scale is None:
                    ratio_scale = torch.cat([valid_ratios, valid_ratios], -1)[:, None]
                ref_in = reference_points.detach()[:, :, None] * ratio_scale        # [B,N,L,4]
                emb = sine_pos_embed(ref_in[:, :, 0, :], self.embed_dim // 2).to(query.dtype)
            qpp = None
            scale_layers = None if idx == 0 else self.query_scale.layers
            if (self.options.decoder_tail and query.is_cuda and not torch.is_grad_enabled() and emb.dtype == torch.bfloat16
                    and ops.query_pos_k256_supported(emb, query, self.ref_point_head.layers, scale_layers)):
                # bf16 inference: both MLPs, their product and query + query_pos in ONE launch (csrc/qpos.hip) instead of four
                # GEMMs + one elementwise launch of the decoder's dependency chain
                query_pos, qpp = ops.query_pos_k256(emb, query, self.ref_point_head.layers, scale_layers)
            else:
                query_pos = self.ref_point_head(emb)
            if idx != 0 and qpp is None:
                if query.is_cuda and not torch.is_grad_enabled() and query.dtype in (torch.float32, torch.bfloat16) \
                        and query_pos.dtype == query.dtype and self.options.decoder_entry:
                    query_pos, qpp = ops.scaled_pos(query_pos, self.query_scale(query), query)      # the product and query + product
                else:
                    query_pos = query_pos * self.query_scale(query)
            query = layer(query=query, query_pos=query_pos, reference_points=ref_in, value=value,
                          spatial_shapes=spatial_shapes, level_start_index=level_start_index,
                          key_padding_mask=key_padding_mask, self_attn_mask=pos_relation,
                          **({} if qpp is None else {"query_plus_pos": qpp}),
                          **({} if values_all is None else
                             {"projected_value": values_all[..., idx * self.embed_dim:(idx + 1) * self.embed_dim]}))
            normed = add_norm(self.norm, query)
            out_class = self.class_head[idx](normed)
            last = idx == self.num_layers - 1
            # bf16 inference: the box head on `normed` (this layer's boxes) and on `query` (the next reference points) with both
            # refinements as ONE kernel (csrc/mlp.hip) instead of 6 GEMMs + 2 launches of the decoder's dependency chain
            fused_box = (query.is_cuda and not torch.is_grad_enabled() and reference_points.dtype == torch.float32
                         and self.options.box_head and ops.box_head_k256_supported(normed, self.bbox_head[idx].layers))
            if fused_box:
                res = ops.box_head_k256(normed, None if last else query, self.bbox_head[idx].layers, reference_points.detach())
                out_coord, next_reference = (res, None) if last else res
            else:
                # boxes stay fp32 whatever the network dtype (a bf16 + fp32 add takes torch's slow mixed-dtype kernel)
                out_coord = refine_boxes(self.bbox_head[idx](normed), reference_points)
            classes.append(out_class)
            coords.append(out_coord)
            if last:
                break
            if not skip_relation:                     # bias for the NEXT layer's self-attention (:369-374)
                src_boxes = tgt_boxes if idx >= 1 else reference_points
                tgt_boxes = out_coord
                if (query.is_cuda and query.dtype == torch.bfloat16 and not torch.is_grad_enabled()
                        and hasattr(self.position_relation_embedding, "deferred") and self.options.rel_fused):
                    # bf16 inference: hand the next layer the recipe; its attention kernel generates the bias (csrc/attn_rel.hip)
                    pos_relation = self.position_relation_embedding.deferred(src_boxes, tgt_boxes, attn_mask)
                else:
                    pos_relation = self.position_relation_embedding(src_boxes, tgt_boxes).flatten(0, 1)
                    if attn_mask is not None:
                        pos_relation.masked_fill_(attn_mask, float("-inf"))
            reference_points = next_reference if fused_box else refine_boxes(self.bbox_head[idx](query), reference_points.detach())
        return torch.stack(classes), torch.stack(coords)


class RelationTransformer(nn.Module):
    """Two-stage transformer (relation_transformer.py:17-160): pyramids -> decoder layer outputs + encoder proposals, plus
    the hybrid (one-to-many) branch and the denoising-query concatenation when training."""

    def __init__(self, encoder: RelationTransformerEncoder, decoder: RelationTransformerDecoder, num_classes: int,
                 num_feature_levels: int = 4, two_stage_num_proposals: int = 900, hybrid_num_proposals: int = 900):
        super().__init__()
        d = encoder.embed_dim
        self.embed_dim, self.num_feature_levels = d, num_feature_levels
        self.two_stage_num_proposals, self.num_classes = two_stage_num_proposals, num_classes
        self.hybrid_num_proposals = hybrid_num_proposals
        self.options = _options.get()
        self.level_embeds = nn.Parameter(torch.empty(num_feature_levels, d))
        self.enc_output = nn.Linear(d, d)
        self.enc_output_norm = nn.LayerNorm(d)
        self.encoder, self.decoder = encoder, decoder
        self.tgt_embed = nn.Embedding(two_stage_num_proposals, d)
        self.encoder_class_head = nn.Linear(d, num_classes)
        self.encoder_bbox_head = MLP(d, d, 4, 3)
        # heads of the hybrid (one-to-many) branch, used in training mode only
        self.hybrid_tgt_embed = nn.Embedding(hybrid_num_proposals, d)
        self.hybrid_class_head = nn.Linear(d, num_classes)
        self.hybrid_bbox_head = MLP(d, d, 4, 3)
        nn.init.normal_(self.level_embeds)
        nn.init.xavier_uniform_(self.enc_output.weight)
        nn.init.zeros_(self.enc_output.bias)
        nn.init.normal_(self.tgt_embed.weight)
        nn.init.normal_(self.hybrid_tgt_embed.weight)
        prior = -math.log((1 - 0.01) / 0.01)
        for head in (self.encoder_class_head, self.hybrid_class_head):
            nn.init.constant_(head.bias, prior)
        for head in (self.encoder_bbox_head, self.hybrid_bbox_head):
            nn.init.zeros_(head.layers[-1].weight)
            nn.init.zeros_(head.layers[-1].bias)

    # ---- pyramid bookkeeping (models/bricks/base_transformer.py:17-81) -------------------------------------
    @staticmethod
    def flatten_levels(levels: Sequence[Tensor]) -> Tensor:
        flat = torch.cat([t.flatten(-2) for t in levels], dim=-1)                   # [B,(C,)S]
        # the reference leaves [B,S,C] as a transposed view (base_transformer.py:17-23); every later `x + pos` on that
        # view is a strided kernel, so pay for one contiguous copy here instead
        return flat.transpose(1, 2).contiguous() if flat.dim() == 3 else flat

    @staticmethod
    def _fast(t: Tensor) -> bool:
        return t.is_cuda and not torch.is_grad_enabled() and t.dtype in (torch.float32, torch.bfloat16)

    # Everything that depends only on the pyramid's level shapes is built once per (shapes, device) and kept: the shape
    # tables, each pixel's centre / level size / level index and the proposal sizes.  Nothing in `forward` then reads a
    # device tensor back or uploads a host one, so the whole eval forward can be captured in a HIP graph (graph.py).
    _geometry_cache: dict = {}

    @classmethod
    def level_geometry(cls, level_hw: Sequence[tuple], device) -> dict:
        key = (tuple(level_hw), str(device))
        geo = cls._geometry_cache.get(key)
        if geo is None:
            shapes = torch.tensor(level_hw, dtype=torch.int64, device=device)
            start = torch.cat((shapes.new_zeros(1), shapes.prod(1).cumsum(0)[:-1]))
            centre, size, index, prop = [], [], [], []
            for lvl, (h, w) in enumerate(level_hw):
                ys, xs = torch.meshgrid(torch.arange(0.5, h + 0.5, device=device), torch.arange(0.5, w + 0.5, device=device),
                                        indexing="ij")
                centre.append(torch.stack((xs.reshape(-1), ys.reshape(-1)), -1))
                size.append(torch.tensor([w, h], dtype=torch.float32, device=device).expand(h * w, 2))
                index.append(torch.full((h * w,), lvl, dtype=torch.int64, device=device))
                prop.append(torch.full((h * w, 2), 0.05 * 2.0 ** lvl, dtype=torch.float32, device=device))
            geo = dict(shapes=shapes, start=start, centre=torch.cat(centre), size=torch.cat(size), index=torch.cat(index),
                       proposal_wh=torch.cat(prop))
            if len(cls._geometry_cache) > 16:
                cls._geometry_cache.clear()
            if torch.device(device).type == "cuda" and not torch.cuda.is_current_stream_capturing():
                torch.cuda.current_stream(device).synchronize()       # built once; other streams (image groups) read it
            cls._geometry_cache[key] = geo
        return geo

    @classmethod
    def level_misc(cls, masks: Sequence[Tensor]):
        geo = cls.level_geometry([tuple(m.shape[-2:]) for m in masks], masks[0].device)
        ratios = []
        for m in masks:
            _, h, w = m.shape
            ratios.append(torch.stack([(~m[:, 0, :]).sum(1).float() / w, (~m[:, :, 0]).sum(1).float() / h], -1))
        return geo, torch.stack(ratios, 1)                                          # valid_ratios [B,L,2] (w,h)

    @staticmethod
    def reference_and_proposals(geo: dict, valid_ratios: Tensor):
        # pixel centre / (valid_ratio of its level * level size): the same quotient as base_transformer.py:57-70
        full = geo["centre"][None] / (valid_ratios[:, geo["index"]] * geo["size"][None])      # [B,S,2]
        reference = full[:, :, None] * valid_ratios[:, None]                        # [B,S,L,2]
        prop = geo["proposal_wh"][None].expand(full.shape[0], -1, -1)
        return reference, torch.cat([full, prop], -1)

    def encoder_output(self, memory: Tensor, proposals: Tensor, padding_mask: Tensor):
        valid = ((proposals > 0.01) & (proposals < 0.99)).all(-1, keepdim=True)
        logit = torch.log(proposals / (1 - proposals))
        logit = logit.masked_fill(padding_mask.unsqueeze(-1) | ~valid, float("inf"))
        keep = (~padding_mask.unsqueeze(-1)) & valid                                 # one pass over memory instead of two
        out = memory * keep.to(memory.dtype)
        return add_norm(self.enc_output_norm, self.enc_output(out)), logit

    def forward(self, multi_level_feats: Sequence[Tensor], multi_level_masks: Sequence[Tensor],
                multi_level_pos_embeds: Sequence[Tensor], noised_label_query: Tensor = None, noised_box_query: Tensor = None,
                attn_mask: Tensor = None):
        """relation_transformer.py:59-160, same arguments and the same 8 results: (layer class logits [Ld,B,N,C], layer
        boxes [Ld,B,N,4], encoder top-k logits, encoder top-k boxes, hybrid layer logits, hybrid layer boxes, hybrid
        encoder logits, hybrid encoder boxes) -- the last four are None unless the module is in training mode.
        ``noised_label_query [B,Ndn,d]`` / ``noised_box_query [B,Ndn,4]`` (logit space) are the denoising queries put in
        front of the matching queries (:120-123), ``attn_mask [Ndn+N, Ndn+N]`` bool their visibility mask."""
        mask = self.flatten_levels(multi_level_masks)
        fusion_buffer = None
        if self._fast(multi_level_feats[0]):
            # inference: one transposing pass per level straight into the token tensors (csrc/glue.hip) -- the features go into
            # the first column block of the encoder's memory-fusion input, the level embedding is added on the way
            d, B = self.embed_dim, multi_level_feats[0].shape[0]
            fusion_buffer = torch.empty(B, mask.shape[1], (self.encoder.num_layers + 1) * d, dtype=multi_level_feats[0].dtype,
                                        device=mask.device)
            feat = ops.tokens_from_levels(multi_level_feats, out=fusion_buffer[..., :d])
            pos = ops.tokens_from_levels(multi_level_pos_embeds, add_vecs=list(self.level_embeds))
        else:
            feat = self.flatten_levels(multi_level_feats)
            pos = self.flatten_levels([p + e.view(1, -1, 1, 1) for p, e in zip(multi_level_pos_embeds, self.level_embeds)])
        fast = self._fast(multi_level_feats[0]) and len(multi_level_masks) <= 8 and self.options.pyramid_points
        if fast:
            # inference: valid ratios, reference points, proposal logits and the validity factor in two launches (csrc/glue.hip)
            # instead of ~40 small ones
            geo = self.level_geometry([tuple(m.shape[-2:]) for m in multi_level_masks], mask.device)
            valid_ratios, reference, out_proposals, keep = ops.pyramid_points(multi_level_masks, mask, feat.dtype)
        else:
            geo, valid_ratios = self.level_misc(multi_level_masks)
            reference, proposals = self.reference_and_proposals(geo, valid_ratios)
        shapes, start = geo["shapes"], geo["start"]

        memory = self.encoder(query=feat, query_pos=pos, query_key_padding_mask=mask, spatial_shapes=shapes,
                              level_start_index=start, reference_points=reference, fusion_buffer=fusion_buffer)

        if fast:
            out_memory = add_norm(self.enc_output_norm, self.enc_output(memory * keep.unsqueeze(-1)))
        else:
            out_memory, out_proposals = self.encoder_output(memory, proposals, mask)
        enc_class, enc_coord = self._top_proposals(out_memory, out_proposals, self.encoder_class_head, self.encoder_bbox_head,
                                                   self.two_stage_num_proposals)
        target = self.tgt_embed.weight.expand(feat.shape[0], -1, -1)
        reference_points = enc_coord.detach()

        hybrid_enc_class = hybrid_enc_coord = hybrid_classes = hybrid_coords = None
        if self.training:                               # one-to-many branch: its own heads, queries and proposal count (:101-115)
            hybrid_enc_class, hybrid_enc_coord = self._top_proposals(out_memory, out_proposals, self.hybrid_class_head,
                                                                     self.hybrid_bbox_head, self.hybrid_num_proposals)
            hybrid_target = self.hybrid_tgt_embed.weight.expand(feat.shape[0], -1, -1)

        if noised_label_query is not None and noised_box_query is not None:
            target = torch.cat([noised_label_query.to(target.dtype), target], 1)
            reference_points = torch.cat([noised_box_query.sigmoid().to(reference_points.dtype), reference_points], 1)

        classes, coords = self.decoder(query=target, value=memory, key_padding_mask=mask, reference_points=reference_points,
                                       spatial_shapes=shapes, level_start_index=start, valid_ratios=valid_ratios,
                                       attn_mask=attn_mask)
        if self.training:                               # same decoder weights, no relation bias, no mask (:136-146)
            hybrid_classes, hybrid_coords = self.decoder(
                query=hybrid_target, value=memory, key_padding_mask=mask, reference_points=hybrid_enc_coord.detach(),
                spatial_shapes=shapes, level_start_index=start, valid_ratios=valid_ratios, skip_relation=True)
        return (classes, coords, enc_class, enc_coord, hybrid_classes, hybrid_coords, hybrid_enc_class, hybrid_enc_coord)

    def _top_proposals(self, out_memory: Tensor, out_proposals: Tensor, class_head: nn.Linear, bbox_head: MLP, k: int):
        """Class logits and sigmoid boxes of the k best-scoring encoder tokens (:86-96 and :101-111)."""
        logits = class_head(out_memory)
        if self._fast(logits):
            # inference: the box head is per token, so it runs on the k selected tokens instead of all S (the reference computes all
            # boxes and gathers, :88-96 -- same values, 3 GEMMs on 900 rows instead of 22,323 per image)
            scores = ops.row_max(logits)
            use_own = self.options.topk and ops.topk_supported(scores, k)
            top = (ops.topk(scores, k)[1] if use_own else torch.topk(scores, k, dim=1)[1]).unsqueeze(-1)
            sel = out_memory.gather(1, top.expand(-1, -1, out_memory.shape[-1]))
            prop = out_proposals.gather(1, top.expand(-1, -1, 4))
            if self.options.box_head and prop.dtype == torch.float32 and ops.box_head_k256_supported(sel, bbox_head.layers):
                boxes = ops.box_head_k256(sel, None, bbox_head.layers, prop, reference_is_logit=True)      # 3 GEMMs + add + sigmoid
            else:
                boxes = (bbox_head(sel).float() + prop).sigmoid()
            return logits.gather(1, top.expand(-1, -1, self.num_classes)), boxes
        boxes = (bbox_head(out_memory).float() + out_proposals).sigmoid()          # fp32 boxes, no mixed-dtype add
        top = torch.topk(logits.max(-1)[0], k, dim=1)[1].unsqueeze(-1)
        return logits.gather(1, top.expand(-1, -1, self.num_classes)), boxes.gather(1, top.expand(-1, -1, 4))


def build_relation_transformer(num_classes=91, embed_dim=256, num_heads=8, d_ffn=2048, num_levels=4, num_points=4,
                               enc_layers=6, dec_layers=6, num_queries=900, hybrid_num_proposals=1500,
                               msda_cls=MultiScaleDeformableAttention, self_attn_cls=RelationSelfAttention,
                               relation_cls=PositionRelationEmbedding) -> RelationTransformer:
    """configs/relation_detr/relation_detr_resnet50_800_1333.py:46-78 (transformer part)."""
    enc = RelationTransformerEncoder([
        RelationTransformerEncoderLayer(embed_dim, d_ffn, num_heads, num_levels, num_points, msda_cls)
        for _ in range(enc_layers)])
    dec = RelationTransformerDecoder([
        RelationTransformerDecoderLayer(embed_dim, d_ffn, num_heads, num_levels, num_points, msda_cls, self_attn_cls)
        for _ in range(dec_layers)], num_classes, relation_cls)
    return RelationTransformer(enc, dec, num_classes, num_levels, num_queries, hybrid_num_proposals)


@torch.no_grad()
def select_detections(logits: Tensor, boxes: Tensor, image_sizes: Tensor, k: int = 300, opts: "_options.Options" = None) -> Tensor:
    """Fixed-shape form of PostProcess (models/bricks/post_process.py:21-44, default config: 300 detections,
    no NMS / score filter): sigmoid -> top-k over N*C -> cxcywh to xyxy -> scale to pixels.
    logits [B,N,C], boxes [B,N,4] cxcywh in [0,1], image_sizes [B,2] (h,w) -> [B,k,6] = (x1,y1,x2,y2,score,label),
    the tensor `dist.gather_detections` all-gathers."""
    B, N, C = logits.shape
    opts = opts or _options.get()
    prob = logits.sigmoid().view(B, -1)
    if logits.is_cuda and opts.topk and ops.topk_supported(prob, k):
        score, idx = ops.topk(prob, k)                  # equal scores by ascending index (csrc/topk.hip); torch: unspecified
        score = score.to(prob.dtype)
    else:
        score, idx = torch.topk(prob, k, dim=1)
    if (logits.is_cuda and score.dtype == torch.float32 and boxes.dtype == torch.float32 and image_sizes.dtype == torch.int64
            and opts.detections_kernel):
        return ops.detections_from_topk(score, idx, boxes, image_sizes, C)       # the rest of this function in one launch
    box_idx = torch.div(idx, C, rounding_mode="trunc")
    label = idx % C
    cx, cy, w, h = boxes.gather(1, box_idx.unsqueeze(-1).expand(-1, -1, 4)).unbind(-1)
    xyxy = torch.stack([cx - 0.5 * w, cy - 0.5 * h, cx + 0.5 * w, cy + 0.5 * h], -1)
    img_h, img_w = image_sizes.to(xyxy.dtype).unbind(1)
    xyxy = xyxy * torch.stack([img_w, img_h, img_w, img_h], 1)[:, None, :]
    return torch.cat([xyxy, score.unsqueeze(-1).to(xyxy.dtype), label.unsqueeze(-1).to(xyxy.dtype)], -1)
