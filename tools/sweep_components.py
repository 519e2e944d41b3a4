#!/usr/bin/env python3
"""Component timing of the sweep MSDA kernel with the development library (make -C relation_detr_amd/csrc dev):
    RDETR_LIB_PATH=relation_detr_amd/librelation_detr_amd_dev.so python3 tools/sweep_components.py [reps]
dbg bits: 1 = no ring fills, 2 = no MFMA steps, 4 = no flagged-sample adds / output store, 8 = every sample the zero sample,
16 = (unused) (results are wrong with any bit set).  MASKS=0,1,2,... selects the runs."""
import ctypes
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from relation_detr_amd import _lib, ops  # noqa: E402


def main():
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 50
    algo = os.environ.get("ALGO", "sweep")
    setdbg = getattr(ctypes.CDLL(_lib.LIB_PATH), "rdetr_dev_set_%s_dbg" % algo)
    dev = torch.device("cuda", 0)
    value, shapes, start, loc, attn, S, L = bench.encoder_kernel_inputs(4, dev, torch.bfloat16)
    vh = value.permute(0, 2, 1, 3).contiguous()
    run = lambda: ops.ms_deform_attn_forward(vh, shapes, start, loc, attn, value_layout="bhsd", algo=algo)
    for _ in range(200):
        run()
    for mask in [int(x) for x in os.environ.get("MASKS", "0,1,2,3,4,7,15,23,31,0").split(",")]:
        setdbg(mask)
        for _ in range(20):
            run()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            run()
        e1.record()
        torch.cuda.synchronize()
        print(f"dbg={mask}: {e0.elapsed_time(e1) / reps * 1e3:7.1f} us", flush=True)
    setdbg(0)


if __name__ == "__main__":
    main()
