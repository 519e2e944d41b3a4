"""relation_detr_amd -- MI355X-native hot path of Relation-DETR.

Multi-scale deformable attention + position-relation attention bias as hand-written gfx950 HIP
kernels behind a C ABI (include/relation_detr_amd.h), with the reference's nn.Module API on top.
"""
from .ms_deform_attn import MultiScaleDeformableAttention
from .relation import PositionRelationEmbedding, PositionRelationEncoder, box_rel_encoding
from .self_attn import RelationSelfAttention
from .transformer import RelationTransformer, build_relation_transformer, select_detections
from .ops import (MultiScaleDeformableAttnFunction, bias_softmax_, ms_deform_attn_backward, ms_deform_attn_forward,
                  ms_deform_attn_forward_fused, relation_bias)

__all__ = [
    "MultiScaleDeformableAttention", "PositionRelationEmbedding", "PositionRelationEncoder", "box_rel_encoding",
    "RelationSelfAttention", "RelationTransformer", "build_relation_transformer", "select_detections",
    "MultiScaleDeformableAttnFunction", "ms_deform_attn_forward",
    "ms_deform_attn_forward_fused", "ms_deform_attn_backward", "relation_bias", "bias_softmax_",
]
