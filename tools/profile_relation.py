#!/usr/bin/env python3
"""Time the relation-bias and bias-softmax kernels at the decoder shape (B=4, N=900 / 300, 8 heads)."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import relation_detr_amd as rd

def timeit(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3

dev = "cuda:0"
g = torch.Generator().manual_seed(0)
for N in (900, 300):
    B = 4
    src = torch.cat([torch.rand(B, N, 2, generator=g), torch.rand(B, N, 2, generator=g) * 0.49 + 0.01], -1).to(dev)
    tgt = torch.cat([torch.rand(B, N, 2, generator=g), torch.rand(B, N, 2, generator=g) * 0.49 + 0.01], -1).to(dev)
    w = ((torch.rand(8, 64, 1, 1, generator=g) - 0.5) * 0.25).to(dev)
    b = ((torch.rand(8, generator=g) - 0.5) * 0.25).to(dev)
    t_rel = timeit(lambda: rd.relation_bias(src, tgt, w, b))
    bias = rd.relation_bias(src, tgt, w, b).flatten(0, 1).contiguous()
    scores = torch.randn(B * 8, N, N, device=dev)
    t_sm = timeit(lambda: rd.bias_softmax_(scores, bias))
    out_mb = B * 8 * N * N * 4 / 1e6
    print(f"N={N}: relation_bias {t_rel:.1f} us ({out_mb / t_rel:.2f} TB/s of output write), "
          f"bias_softmax {t_sm:.1f} us ({3 * out_mb / t_sm:.2f} TB/s)")
