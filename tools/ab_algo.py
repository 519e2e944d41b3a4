#!/usr/bin/env python3
"""Time ops.ms_deform_attn_forward (bf16, encoder shape, B=4) under the RDETR_MSDA_ALGO of the environment.
    RDETR_MSDA_ALGO=tile2d python3 tools/ab_algo.py [reps]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from relation_detr_amd import ops  # noqa: E402
from tools.ab_msda import timed  # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
dev = torch.device("cuda", 0)
for dtype in (torch.bfloat16, torch.float32):
    value, shapes, start, loc, attn, S, L = bench.encoder_kernel_inputs(4, dev, dtype)
    t = timed(lambda: ops.ms_deform_attn_forward(value, shapes, start, loc, attn), reps)
    print(f"ALGO={os.environ.get('RDETR_MSDA_ALGO', 'qrun'):7s} {str(dtype):15s} {t*1e6:7.1f} us", flush=True)
