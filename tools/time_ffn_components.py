#!/usr/bin/env python3
"""Component builds of the fused feed-forward kernel (development library, make -C relation_detr_amd/csrc dev):
    RDETR_LIB_PATH=relation_detr_amd/librelation_detr_amd_dev.so python3 tools/time_ffn_components.py
dbg bits (csrc/ffn.hip): 1 = no weight stream, 2 = no per-chunk workgroup barrier, 4 / 8 = no GEMM 2 / GEMM 1 (wrong results with
any bit set).  dbg = 2 is the UPPER BOUND of every replacement of the per-chunk barrier (LDS flags, split barriers): the same
kernel with no synchronisation at all between the waves and the weight stream."""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from relation_detr_amd import _lib, ops  # noqa: E402
from tools.time_linear import timed  # noqa: E402

setdbg = ctypes.CDLL(_lib.LIB_PATH).rdetr_dev_set_ffn_dbg
F = 2048
w1 = (torch.randn(F, 256, device="cuda") * 0.05).bfloat16()
b1 = torch.randn(F, device="cuda").bfloat16()
w2 = (torch.randn(256, F, device="cuda") * 0.02).bfloat16()
b2 = torch.randn(256, device="cuda").bfloat16()
for M in (44646, 89292):
    x = torch.randn(M, 256, device="cuda").bfloat16()
    for rnd in range(2):
        for mask in (0, 2, 1, 3, 0):
            setdbg(mask)
            a = timed(lambda: ops.ffn_k256(x, w1, b1, w2, b2))
            print(f"M={M:6d} round {rnd} dbg={mask}: {a*1e6:6.1f} us", flush=True)
setdbg(0)
