#!/usr/bin/env python3
"""Run only the LDS-window MSDA kernel at the encoder shape of BASELINE.json configs[1] (B=4, S=Nq=22,323, 4 levels), so that
rocprofv3 traces / PMC passes of it stay small.
    rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/x -- python3 tools/profile_win.py [bhsd|bshd] [reps] [window|direct|auto] [r50|focalnet]
    rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES ... --output-format csv -d gpurun_out/y -- python3 tools/profile_win.py"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from relation_detr_amd import ops  # noqa: E402


def main():
    lay = sys.argv[1] if len(sys.argv) > 1 else "bhsd"
    reps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
    algo = sys.argv[3] if len(sys.argv) > 3 else "window"
    dev = torch.device("cuda", 0)
    if os.environ.get("RDETR_DEV_IDENTITY_ORDER") == "1":                  # development library only: blocks in query order
        import ctypes
        from relation_detr_amd import _lib
        ctypes.CDLL(_lib.LIB_PATH).rdetr_dev_set_msda_identity_order(1)
    if os.environ.get("RDETR_DEV_HEAD_GROUP_LOG2"):                         # development library only: heads per XCD-local group
        import ctypes
        from relation_detr_amd import _lib
        ctypes.CDLL(_lib.LIB_PATH).rdetr_dev_set_msda_head_group_log2(int(os.environ["RDETR_DEV_HEAD_GROUP_LOG2"]))
    if os.environ.get("RDETR_DEV_RES_PLANE_MAJOR"):                         # development library only: resident kernel's plane -> XCD map
        import ctypes
        from relation_detr_amd import _lib
        ctypes.CDLL(_lib.LIB_PATH).rdetr_dev_set_res_plane_major(int(os.environ["RDETR_DEV_RES_PLANE_MAJOR"]))
    if os.environ.get("RDETR_DEV_RES_MAX_TEAMS"):                           # development library only: planes of an XCD in flight
        import ctypes
        from relation_detr_amd import _lib
        ctypes.CDLL(_lib.LIB_PATH).rdetr_dev_set_res_max_teams(int(os.environ["RDETR_DEV_RES_MAX_TEAMS"]))
    if os.environ.get("RDETR_DEV_RES_TILED"):                               # development library only: resident kernel's query order
        import ctypes
        from relation_detr_amd import _lib
        ctypes.CDLL(_lib.LIB_PATH).rdetr_dev_set_res_tiled(int(os.environ["RDETR_DEV_RES_TILED"]))
    cfg = bench.CONFIGS[sys.argv[4]] if len(sys.argv) > 4 else bench.CONFIGS["r50"]          # [config]: r50 (B = 4) | focalnet (B = 2)
    nb = int(os.environ["RDETR_PROFILE_B"]) if os.environ.get("RDETR_PROFILE_B") else cfg["batch"]
    value, shapes, start, loc, attn, S, L = bench.encoder_kernel_inputs(nb, dev, torch.bfloat16, cfg["shapes"])
    v = value.permute(0, 2, 1, 3).contiguous() if lay == "bhsd" else value
    if os.environ.get("RDETR_PROFILE_FUSED") == "1":                       # the fused-producer form (what the stack launches)
        from tools.ab_head_group import fused_inputs
        fi = fused_inputs(cfg["batch"] if not os.environ.get("RDETR_PROFILE_B") else int(os.environ["RDETR_PROFILE_B"]), dev, cfg["shapes"])
        for _ in range(reps):
            ops.ms_deform_attn_forward_fused(fi[0], fi[1], fi[2], fi[3], fi[4], fi[5], value_layout="bhsd", algo=algo)
        torch.cuda.synchronize()
        return
    for _ in range(reps):
        ops.ms_deform_attn_forward(v, shapes, start, loc, attn, value_layout=lay, algo=algo)
    torch.cuda.synchronize()


if __name__ == "__main__":
    main()
