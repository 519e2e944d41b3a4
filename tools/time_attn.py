#!/usr/bin/env python3
"""GPU time of the decoder self-attention at the decoder's size, by graph replay: the relation-bias kernel + the attention
kernel that reads the bias (csrc/relation.hip + csrc/attn.hip) against the attention kernel that generates the bias itself
(csrc/attn_rel.hip)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from relation_detr_amd import ops  # noqa: E402
from tools.time_linear import timed  # noqa: E402

H = 8
for B, N in ((4, 900), (2, 900), (1, 900), (2, 300)):
    q = torch.randn(B, N, 256, device="cuda").bfloat16()
    k = torch.randn(B, N, 256, device="cuda").bfloat16()
    v = torch.randn(B, N, 256, device="cuda").bfloat16()
    src = torch.cat([torch.rand(B, N, 2, device="cuda"), torch.rand(B, N, 2, device="cuda") * 0.49 + 0.01], -1)
    tgt = torch.cat([torch.rand(B, N, 2, device="cuda"), torch.rand(B, N, 2, device="cuda") * 0.49 + 0.01], -1)
    w = (torch.rand(8, 64, device="cuda") - 0.5) * 0.25
    b = torch.zeros(8, device="cuda")
    bias = ops.relation_bias(src, tgt, w, b).flatten(0, 1)
    t_bias = timed(lambda: ops.relation_bias(src, tgt, w, b))
    t_attn = timed(lambda: ops.relation_attention(q, k, v, H, bias))
    t_both = timed(lambda: ops.relation_attention(q, k, v, H, ops.relation_bias(src, tgt, w, b).flatten(0, 1)))
    t_gen = timed(lambda: ops.relation_attention_boxes(q, k, v, H, src, tgt, w, b))
    print(f"B={B} N={N}: relation_bias {t_bias*1e6:6.1f} us + relation_attention {t_attn*1e6:6.1f} us = {t_both*1e6:6.1f} us in "
          f"sequence;  relation_attention_boxes {t_gen*1e6:6.1f} us", flush=True)
