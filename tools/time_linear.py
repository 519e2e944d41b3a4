#!/usr/bin/env python3
"""ops.linear_k256 (csrc/linear.hip) against the library GEMM behind F.linear on the path's K = 256 layers."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from relation_detr_amd import ops  # noqa: E402


def timed(fn, reps=20):
    """GPU time per call: `reps` calls captured into one graph (no host enqueue time in the figure)."""
    fn()
    torch.cuda.synchronize()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        fn()
    torch.cuda.current_stream().wait_stream(side)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(reps):
            fn()
    g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / (5 * reps) * 1e-3


if __name__ == "__main__":
    for M in (44646, 89292, 3600):
        for N, relu in ((256, False), (384, False), (2048, True), (512, False)):
            x = torch.randn(M, 256, device="cuda").bfloat16()
            w = (torch.randn(N, 256, device="cuda") * 0.05).bfloat16()
            b = torch.randn(N, device="cuda").bfloat16()
            a = timed(lambda: ops.linear_k256(x, w, b, relu=relu))
            if relu:
                t = timed(lambda: torch._addmm_activation(b, x, w.t()))
            else:
                t = timed(lambda: torch.nn.functional.linear(x, w, b))
            nbytes = (M * 256 + M * N) * 2
            print(f"M={M:6d} N={N:4d} relu={int(relu)}: linear_k256 {a*1e6:6.1f} us ({nbytes/a/1e12:4.2f} TB/s, {2.0*M*N*256/a/1e12:5.0f} TF)   "
                  f"library {t*1e6:6.1f} us", flush=True)
