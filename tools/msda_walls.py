#!/usr/bin/env python3
"""Where the bf16 head-major gather kernel's time goes: the same launch (B=4 encoder shape of BASELINE.json configs[1]) with
sampling locations that take parts of the memory system out of the picture.
    real     the bench's locations (reference point + offsets of up to ~4 px)
    outside  every location outside the image: all corners fail the range check -> no gather request leaves the CU
             (set-up, staging, the 64 gather instructions per wave and the streaming of locations / weights / output remain)
    onepixel every location = the centre of its level: all gathers hit the same few lines in L1
    centre   every query samples its own pixel (no offsets): perfect locality, one line per query and level
    python3 tools/msda_walls.py [algo] [B]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from relation_detr_amd import ops  # noqa: E402
from tools.ab_msda import timed  # noqa: E402


def main():
    algo = sys.argv[1] if len(sys.argv) > 1 else "direct"
    B = int(sys.argv[2]) if len(sys.argv) > 2 else 4
    dev = torch.device("cuda", 0)
    value, shapes, start, loc, attn, S, L = bench.encoder_kernel_inputs(B, dev, torch.bfloat16)
    vh = value.permute(0, 2, 1, 3).contiguous()
    sh = shapes.tolist()
    ref = torch.cat([torch.stack(torch.meshgrid((torch.arange(h, device=dev) + 0.5) / h, (torch.arange(w, device=dev) + 0.5) / w,
                                                indexing="ij")[::-1], -1).reshape(-1, 2) for h, w in sh], 0)
    variants = {
        "real": loc,
        "outside": torch.full_like(loc, -5.0),
        "onepixel": torch.full_like(loc, 0.5),
        "centre": ref.view(1, S, 1, 1, 1, 2).expand_as(loc).contiguous(),
    }
    for name, lc in variants.items():
        ts = [timed(lambda: ops.ms_deform_attn_forward(vh, shapes, start, lc, attn, value_layout="bhsd", algo=algo), 20) for _ in range(3)]
        print(f"{algo:10s} {name:9s}: min {min(ts):7.1f} us  median {sorted(ts)[1]:7.1f} us")


if __name__ == "__main__":
    main()
