#!/usr/bin/env python3
"""Component timing of the LDS-window MSDA kernel with the development library (make -C relation_detr_amd/csrc dev):
masks switch parts of the kernel off (WRONG results, timing only).
    RDETR_LIB_PATH=relation_detr_amd/librelation_detr_amd_dev.so python3 tools/win_components.py [reps]
mask bits: 2 = no window fills, 4 = no passes, 8 = no output store, 16 = no location loads, 32 = no MFMA loop"""
import ctypes
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from relation_detr_amd import _lib, ops  # noqa: E402


def main():
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    lib = _lib.load()
    setdbg = ctypes.CDLL(_lib.LIB_PATH).rdetr_dev_set_win_dbg
    dev = torch.device("cuda", 0)
    value, shapes, start, loc, attn, S, L = bench.encoder_kernel_inputs(4, dev, torch.bfloat16)
    vh = value.permute(0, 2, 1, 3).contiguous()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for name, mask in [("everything", 0), ("no fills", 2), ("no MFMA loop", 32), ("no fills, no MFMA loop", 34), ("no passes", 4),
                       ("no passes, no fills", 6), ("no passes, no fills, no loads", 22), ("skeleton (no passes/fills/loads/store)", 30),
                       ("no location loads", 16), ("no store", 8), ("prologue only", 94), ("skeleton, no tables", 158), ("skeleton, no barriers", 30 + 256), ("skeleton, no k-loop", 30 + 512), ("skeleton, no query_of", 30 + 1024), ("skeleton, none of the three", 30 + 256 + 512 + 1024)]:
        setdbg(mask)
        for lay, v in (("bhsd", vh), ("bshd", value)):
            f = lambda: ops.ms_deform_attn_forward(v, shapes, start, loc, attn, value_layout=lay, algo="window")
            for _ in range(3):
                f()
            torch.cuda.synchronize()
            e0.record()
            for _ in range(reps):
                f()
            e1.record()
            torch.cuda.synchronize()
            print(f"{name:42s} {lay}: {e0.elapsed_time(e1) / reps * 1e3:7.1f} us", flush=True)
    setdbg(0)


if __name__ == "__main__":
    main()
