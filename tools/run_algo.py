#!/usr/bin/env python3
"""Launch one MSDA kernel variant a few times at the encoder shape of BASELINE.json configs[1] (for rocprofv3 passes).
    python3 tools/run_algo.py <direct|window|tile|sweep> [reps] [B] [layout]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from relation_detr_amd import ops  # noqa: E402

algo = sys.argv[1] if len(sys.argv) > 1 else "sweep"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 6
B = int(sys.argv[3]) if len(sys.argv) > 3 else 4
layout = sys.argv[4] if len(sys.argv) > 4 else "bhsd"
dev = torch.device("cuda", 0)
value, shapes, start, loc, attn, S, L = bench.encoder_kernel_inputs(B, dev, torch.bfloat16)
if layout == "bhsd":
    value = value.permute(0, 2, 1, 3).contiguous()
for _ in range(reps):
    ops.ms_deform_attn_forward(value, shapes, start, loc, attn, value_layout=layout, algo=algo)
torch.cuda.synchronize()
print("done", algo, reps)
