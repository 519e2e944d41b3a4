#!/usr/bin/env python3
"""Time the fused add + LayerNorm on the encoder's row count (B=4: 89,292 rows x 256 channels)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from relation_detr_amd import ops  # noqa: E402
from tools.ab_msda import timed  # noqa: E402

for dtype in (torch.bfloat16, torch.float32):
    x = torch.randn(4, 22323, 256, device="cuda").to(dtype)
    r = torch.randn_like(x)
    w, b = torch.randn(256, device="cuda").to(dtype), torch.randn(256, device="cuda").to(dtype)
    t = timed(lambda: ops.add_layer_norm(x, r, w, b, 1e-5), 50)
    nbytes = 3 * x.numel() * x.element_size()
    print(f"add_layer_norm {str(dtype):15s} {t*1e6:6.1f} us  {nbytes/t/1e12:.2f} TB/s")
