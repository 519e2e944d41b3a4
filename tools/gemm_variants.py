#!/usr/bin/env python3
"""Library-GEMM variants for the path's big dense layers (bf16): nn.Linear's NT form vs a pre-transposed weight (NN form),
with and without TunableOp.  Prints us and TFLOP/s."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tools.ab_msda import timed  # noqa: E402

dev = "cuda"
for M in (44646, 89292):
    for K, N in ((256, 2048), (2048, 256), (256, 256), (256, 384), (1792, 256)):
        x = torch.randn(M, K, device=dev).bfloat16()
        w = (torch.randn(N, K, device=dev) * 0.05).bfloat16()
        b = torch.randn(N, device=dev).bfloat16()
        wt = w.t().contiguous()
        fl = 2.0 * M * K * N
        res = {}
        res["linear(NT)"] = timed(lambda: torch.nn.functional.linear(x, w, b), 30)
        res["addmm(NN)"] = timed(lambda: torch.addmm(b, x, wt), 30)
        if N == 2048:
            res["addmm_act(NT)"] = timed(lambda: torch._addmm_activation(b, x, w.t()), 30)
            res["addmm_act(NN)"] = timed(lambda: torch._addmm_activation(b, x, wt), 30)
        print(f"M={M} K={K} N={N}: " + "  ".join(f"{k} {v*1e6:6.1f}us ({fl/v/1e12:5.0f} TF)" for k, v in res.items()), flush=True)
