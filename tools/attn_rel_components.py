#!/usr/bin/env python3
"""Component timing of the generated-bias attention kernel (csrc/attn_rel.hip) with the development library:
    make -C relation_detr_amd/csrc dev && RDETR_LIB_PATH=relation_detr_amd/librelation_detr_amd_dev.so python3 tools/attn_rel_components.py
mask: 1 = feature waves idle, 2 = attention waves idle (WRONG results, timing only)."""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from relation_detr_amd import _lib, ops  # noqa: E402
from tools.time_linear import timed  # noqa: E402

_lib.load()
setdbg = ctypes.CDLL(_lib.LIB_PATH).rdetr_dev_set_attn_rel_dbg
B, N, H = 2, 900, 8
q, k, v = (torch.randn(B, N, 256, device="cuda").bfloat16() for _ in range(3))
src = torch.cat([torch.rand(B, N, 2, device="cuda"), torch.rand(B, N, 2, device="cuda") * 0.49 + 0.01], -1)
w, b = (torch.rand(8, 64, device="cuda") - 0.5) * 0.25, torch.zeros(8, device="cuda")
for name, mask in (("everything", 0), ("attention waves only", 1), ("feature waves only", 2), ("prologue + barriers", 3)):
    setdbg(mask)
    t = timed(lambda: ops.relation_attention_boxes(q, k, v, H, src, src, w, b))
    print(f"B={B} N={N} {name:24s} {t*1e6:6.1f} us", flush=True)
setdbg(0)
