"""CPU stand-ins for the three hot-path modules, computing with the oracle (TEST INFRASTRUCTURE ONLY).

Same parameters / state_dict keys as the product modules (they subclass them for exactly that), but `forward`
runs oracle/torch_ref.py on the host.  Used (a) by the CPU tests to check the transformer harness glue against the
reference's golden output without a GPU and (b) by bench.py's cpu_baseline leg.  Never imported by the product.
"""
import torch

from relation_detr_amd.ms_deform_attn import MultiScaleDeformableAttention
from relation_detr_amd.relation import PositionRelationEmbedding
from relation_detr_amd.self_attn import RelationSelfAttention

from . import torch_ref


class OracleMSDA(MultiScaleDeformableAttention):
    def forward(self, query, reference_points, value, spatial_shapes, level_start_index, key_padding_mask):
        return torch_ref.msda_module_forward(dict(self.named_parameters()), query, reference_points, value,
                                             spatial_shapes, level_start_index, key_padding_mask, self.num_heads,
                                             self.num_levels, self.num_points)


class OracleSelfAttention(RelationSelfAttention):
    def forward(self, query, key, value, attn_mask=None, need_weights=False, key_padding_mask=None):
        out = torch_ref.self_attn_with_bias(query, key, value, self.in_proj_weight, self.in_proj_bias,
                                            self.out_proj.weight, self.out_proj.bias, attn_mask, self.num_heads)
        return out, None


class OracleRelation(PositionRelationEmbedding):
    def forward(self, src_boxes, tgt_boxes=None):
        conv = self.pos_proj[0]                 # boxes carry no gradient (relation_transformer.py:527-529), the projection does
        return torch_ref.relation_bias(src_boxes.detach(), None if tgt_boxes is None else tgt_boxes.detach(), conv.weight,
                                       conv.bias, self.num_pos_feats, self.temperature, self.scale)
