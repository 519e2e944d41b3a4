#!/usr/bin/env python3
"""ops.linear_ln_k256 (output_proj + residual + LayerNorm) against the library GEMM + ops.add_layer_norm, GPU time by graph replay."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from relation_detr_amd import ops  # noqa: E402
from tools.time_linear import timed  # noqa: E402

w = (torch.randn(256, 256, device="cuda") * 0.05).bfloat16()
b = torch.randn(256, device="cuda").bfloat16()
g, be = torch.randn(256, device="cuda").bfloat16(), torch.randn(256, device="cuda").bfloat16()
for M in (44646, 89292, 3600):
    x = torch.randn(M, 256, device="cuda").bfloat16()
    r = torch.randn(M, 256, device="cuda").bfloat16()
    a = timed(lambda: ops.linear_ln_k256(x, w, b, r, g, be))
    t = timed(lambda: ops.add_layer_norm(r, torch.nn.functional.linear(x, w, b), g, be))
    print(f"M={M:6d}: linear_ln_k256 {a*1e6:6.1f} us   library GEMM + add_layer_norm {t*1e6:6.1f} us", flush=True)
