#!/usr/bin/env python3
"""Component timing of the resident-levels MSDA kernel (development library, wrong results by construction): parts switched off.
    RDETR_LIB_PATH=relation_detr_amd/librelation_detr_amd_dev.so python3 tools/res_components.py"""
import ctypes
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import relation_detr_amd as rd  # noqa: E402
from relation_detr_amd import _lib  # noqa: E402

NAMES = {0: "all", 1: "fine points from the LDS zero row (no texture-path gathers)", 2: "coarse points from the zero row (no bank conflicts)",
         3: "both from the zero row", 4: "no set-up arithmetic", 7: "zero rows + no set-up", 8: "no matrix-core steps / re-pairing",
         9: "no texture-path gathers, no matrix-core steps", 11: "zero rows, no matrix-core steps", 12: "no set-up, no matrix-core steps",
         15: "nothing but the skeleton (inputs, staging, row reads from the zero row, store)"}


def main():
    dev = torch.device("cuda", 0)
    lib = ctypes.CDLL(_lib.LIB_PATH)
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 4
    value, shapes, start, loc, attn, S, L = bench.encoder_kernel_inputs(B, dev, torch.bfloat16)
    vh = value.permute(0, 2, 1, 3).contiguous()
    fn = lambda: rd.ops.ms_deform_attn_forward(vh, shapes, start, loc, attn, value_layout="bhsd", algo="resident")
    order = (0, 1, 2, 3, 4, 8, 9, 7, 11, 12, 15, 0)
    if len(sys.argv) > 2 and sys.argv[2] == "noperm":
        NAMES.update({128: "no v_perm re-pairing at all (wrong operands)", 256: "no v_perm re-pairing for the coarse (LDS) points"})
        order = (0, 256, 128, 0, 256, 128, 0, 256, 128)
    if len(sys.argv) > 2 and sys.argv[2] == "pipe":
        NAMES.update({512: "8 waves, explicit pipeline (weights up front, coarse rows one step ahead)", 64: "8 waves, compiler's schedule, fine rows 1 step ahead"})
        NAMES[0] = "12 waves (product)"
        order = (0, 512, 64, 0, 512, 64, 0, 512, 64)
    if len(sys.argv) > 2 and sys.argv[2] == "ahead":
        NAMES.update({64: "8 waves, fine rows 1 step ahead", 16: "8 waves, 3 steps ahead", 32: "8 waves, 4 steps ahead"})
        lib.rdetr_dev_set_res_waves(8)
        NAMES[0] = "8 waves, 2 steps ahead"
        order = (0, 64, 16, 32, 0, 64, 16, 32)
    for dbg in order:
        lib.rdetr_dev_set_res_dbg(dbg)
        for _ in range(80):
            fn()
        t = bench._timed_launches(fn, 40)
        print(f"B={B} dbg {dbg:2d} {t * 1e6:8.1f} us   {NAMES[dbg]}", flush=True)
    lib.rdetr_dev_set_res_dbg(0)


if __name__ == "__main__":
    main()
