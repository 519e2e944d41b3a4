#!/usr/bin/env python3
"""Diagnostic: cost of the fine vs the coarse pyramid levels in the MSDA gather (B=4, 22,323 queries per image).
Times rd.ms_deform_attn_forward on (a) the full R50 pyramid, (b) levels 0-1 only, (c) levels 2-3 only."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import relation_detr_amd as rd

def run(shapes_list, Nq, B, dtype, reps=20):
    dev = "cuda:0"
    g = torch.Generator().manual_seed(1)
    shapes = torch.tensor(shapes_list, dtype=torch.int64)
    areas = shapes[:, 0] * shapes[:, 1]
    start = torch.cat([areas.new_zeros(1), areas.cumsum(0)[:-1]])
    S, L = int(areas.sum()), len(shapes_list)
    value = torch.randn(B, S, 8, 32, generator=g).to(dev).to(dtype)
    # queries laid out like the R50 level-0 grid (raster order), offsets N(0, k px) in each level's own pixels
    n0 = 100 * 168
    idx = torch.arange(Nq) % n0
    ref = torch.stack([((idx % 168) + 0.5) / 168, ((idx // 168) + 0.5) / 100], -1)
    wh = shapes.flip(-1).float()
    k = torch.arange(1, 5, dtype=torch.float32).view(1, 1, 1, 1, 4, 1)
    off = torch.randn(B, Nq, 8, L, 4, 2, generator=g) * k / wh.view(1, 1, 1, L, 1, 2)
    loc = (ref[None, :, None, None, None, :] + off).contiguous().to(dev)
    attn = torch.softmax(torch.randn(B, Nq, 8, L * 4, generator=g), -1).view(B, Nq, 8, L, 4).contiguous().to(dev)
    shapes, start = shapes.to(dev), start.to(dev)
    for _ in range(3):
        rd.ms_deform_attn_forward(value, shapes, start, loc, attn, 64)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        rd.ms_deform_attn_forward(value, shapes, start, loc, attn, 64)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3

for dt in (torch.float32, torch.bfloat16):
    full = run([(100, 168), (50, 84), (25, 42), (13, 21)], 22323, 4, dt)
    fine = run([(100, 168), (50, 84)], 22323, 4, dt)
    coarse = run([(25, 42), (13, 21)], 22323, 4, dt)
    print(f"{dt}: full 4 levels {full:.0f} us | levels 0-1 only {fine:.0f} us | levels 2-3 only {coarse:.0f} us")
