#!/usr/bin/env python3
"""Where a wave of the LDS-window MSDA kernel spends its cycles (development library, s_memtime stamps summed per wave).
    RDETR_LIB_PATH=relation_detr_amd/librelation_detr_amd_dev.so python3 tools/win_stamps.py"""
import ctypes
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from relation_detr_amd import _lib, ops  # noqa: E402

SEG = ["top: loads + fill issue", "gather (gather-first half)", "set-up", "gather (set-up-first half)", "store + stage",
       "DMA wait", "barrier wait", "-"]


def main():
    _lib.load()
    dll = ctypes.CDLL(_lib.LIB_PATH)
    dev = torch.device("cuda", 0)
    value, shapes, start, loc, attn, S, L = bench.encoder_kernel_inputs(4, dev, torch.bfloat16)
    vh = value.permute(0, 2, 1, 3).contiguous()
    buf = torch.zeros(256 * 16 * 8, dtype=torch.int64, device=dev)
    dll.rdetr_dev_set_win_stamps(ctypes.c_void_p(buf.data_ptr()))
    for _ in range(3):
        ops.ms_deform_attn_forward(vh, shapes, start, loc, attn, value_layout="bhsd", algo="window")
    torch.cuda.synchronize()
    st = buf.view(256, 16, 8).double().cpu()
    dll.rdetr_dev_set_win_stamps(ctypes.c_void_p(0))
    for name, sel in (("gather-first waves 0..7", st[:, :8]), ("set-up-first waves 8..15", st[:, 8:]), ("row waves 0..11", st[:, :12]),
                      ("coarse waves 12..15", st[:, 12:])):
        tot = sel.sum(-1).mean().item()
        print(f"{name}: {tot:10.0f} cycles per wave in the tile loop (clock counts at 100 MHz x ? -- shares matter)")
        for i, n in enumerate(SEG[:7]):
            v = sel[..., i].mean().item()
            print(f"    {n:30s} {v:10.0f}  {100 * v / tot:5.1f} %")


if __name__ == "__main__":
    main()
