#!/usr/bin/env python3
"""A/B timing of the two bf16 MSDA forward strategies on the encoder shape of BASELINE.json configs[1]
(B=4, S=Nq=22,323, 4 levels): "tiled" (csrc/msda_tile.hip) vs "direct" (csrc/msda_fwd.hip), same inputs
(bench.encoder_kernel_inputs), device events on the launch stream.

    python3 tools/ab_msda.py [reps] [B]            RDETR_BENCH_SPREAD=x scales the offset spread
"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from relation_detr_amd import ops  # noqa: E402


def timed(fn, reps):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e-3


def main():
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    B = int(sys.argv[2]) if len(sys.argv) > 2 else 4
    dev = torch.device("cuda", 0)
    value, shapes, start, loc, attn, S, L = bench.encoder_kernel_inputs(B, dev, torch.bfloat16)
    alg = bench.msda_algorithmic_bytes(B, S, S, L, 4, 8, 32, 2)
    outs = {}
    for strat in ("direct", "tiled"):
        t = timed(lambda: ops.ms_deform_attn_forward_strategy(strat, value, shapes, start, loc, attn), reps)
        outs[strat] = ops.ms_deform_attn_forward_strategy(strat, value, shapes, start, loc, attn).float()
        print(f"{strat:7s} B={B}: {t*1e6:7.1f} us/launch  {alg/t/1e9:7.0f} GB/s algorithmic  ({alg/t/8e12*100:.1f}% of 8 TB/s)", flush=True)
    d = (outs["tiled"] - outs["direct"]).abs()
    print(f"max |tiled - direct| = {d.max().item():.4g}  (|direct| max {outs['direct'].abs().max().item():.3g})")


if __name__ == "__main__":
    main()
