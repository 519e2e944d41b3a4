#!/usr/bin/env python3
"""ops.ffn_k256 (csrc/ffn.hip) against the two library GEMMs of the unfused FFN, GPU time by graph replay."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from relation_detr_amd import ops  # noqa: E402
from tools.time_linear import timed  # noqa: E402

F = 2048
w1 = (torch.randn(F, 256, device="cuda") * 0.05).bfloat16()
b1 = torch.randn(F, device="cuda").bfloat16()
w2 = (torch.randn(256, F, device="cuda") * 0.02).bfloat16()
b2 = torch.randn(256, device="cuda").bfloat16()
for M in (44646, 89292, 3600):
    x = torch.randn(M, 256, device="cuda").bfloat16()
    a = timed(lambda: ops.ffn_k256(x, w1, b1, w2, b2))
    t = timed(lambda: torch.nn.functional.linear(torch._addmm_activation(b1, x, w1.t()), w2, b2))
    fl = 4.0 * M * 256 * F
    print(f"M={M:6d} F={F}: ffn_k256 {a*1e6:6.1f} us ({fl/a/1e12:5.0f} TF)   library (2 GEMMs) {t*1e6:6.1f} us ({fl/t/1e12:5.0f} TF)", flush=True)
