#!/usr/bin/env python3
"""A few launches of the fused FFN and of output_proj + LayerNorm at one image group's size, for rocprofv3 passes:
    rocprofv3 --kernel-trace --pmc SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES --output-format csv -d out -- python3 tools/profile_ffn.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from relation_detr_amd import ops  # noqa: E402

M, F = 44646, 2048
g = torch.Generator().manual_seed(0)
x = torch.randn(M, 256, generator=g).cuda().bfloat16()
r = torch.randn(M, 256, generator=g).cuda().bfloat16()
w1 = (torch.randn(F, 256, generator=g) * 0.05).cuda().bfloat16()
b1 = torch.randn(F, generator=g).cuda().bfloat16()
w2 = (torch.randn(256, F, generator=g) * 0.02).cuda().bfloat16()
b2 = torch.randn(256, generator=g).cuda().bfloat16()
wo = (torch.randn(256, 256, generator=g) * 0.05).cuda().bfloat16()
gm, bt = torch.randn(256, generator=g).cuda().bfloat16(), torch.randn(256, generator=g).cuda().bfloat16()
for _ in range(int(sys.argv[1]) if len(sys.argv) > 1 else 6):
    ops.ffn_k256(x, w1, b1, w2, b2)
    ops.linear_ln_k256(x, wo, b2, r, gm, bt)
torch.cuda.synchronize()
