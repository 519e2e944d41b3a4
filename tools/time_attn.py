#!/usr/bin/env python3
"""GPU time of the fused decoder self-attention (csrc/attn.hip) at the decoder's size, by graph replay."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from relation_detr_amd import ops  # noqa: E402
from tools.time_linear import timed  # noqa: E402

for B in (4, 2):
    N, H = 900, 8
    q = torch.randn(B, N, 256, device="cuda").bfloat16()
    k = torch.randn(B, N, 256, device="cuda").bfloat16()
    v = torch.randn(B, N, 256, device="cuda").bfloat16()
    bias = torch.randn(B * H, N, N, device="cuda").relu()
    t = timed(lambda: ops.relation_attention(q, k, v, H, bias))
    print(f"B={B} N={N}: relation_attention {t*1e6:6.1f} us  ({bias.numel()*4/t/1e12:.2f} TB/s of bias)", flush=True)
