#!/usr/bin/env python3
"""GPU time of the relation-bias kernel (csrc/relation.hip) at the decoder's size, by graph replay."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from relation_detr_amd import ops  # noqa: E402
from tools.time_linear import timed  # noqa: E402

w = torch.randn(8, 64, device="cuda") * 0.1
bb = torch.randn(8, device="cuda") * 0.1
for B in (4, 2, 1):
    boxes = torch.rand(B, 900, 4, device="cuda") * 0.5 + 0.1
    t = timed(lambda: ops.relation_bias(boxes, boxes, w, bb))
    print(f"B={B} N=900: relation_bias {t*1e6:6.1f} us", flush=True)
