#!/usr/bin/env python3
"""Same-process A/B of the bf16 MSDA kernels at the encoder shape of BASELINE.json configs[1]
(B=4, S=Nq=22,323, 4 levels): "direct" (csrc/msda_fwd.hip) vs "window" (csrc/msda_win.hip), value in the reference
operator's layout [B,S,H,D] ("bshd") and head-major [B,H,S,D] ("bhsd"), same inputs, interleaved rounds.
    python3 tools/ab_msda.py [reps] [rounds] [B]"""
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
    return e0.elapsed_time(e1) / reps * 1e3


def main():
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 3
    B = int(sys.argv[3]) if len(sys.argv) > 3 else 4
    dev = torch.device("cuda", 0)
    value, shapes, start, loc, attn, S, L = bench.encoder_kernel_inputs(B, dev, torch.bfloat16)
    vh = value.permute(0, 2, 1, 3).contiguous()
    alg = bench.msda_algorithmic_bytes(B, S, S, L, 4, 8, 32, 2)
    arms = {
        "direct bshd": lambda: ops.ms_deform_attn_forward(value, shapes, start, loc, attn, algo="direct"),
        "direct bhsd": lambda: ops.ms_deform_attn_forward(vh, shapes, start, loc, attn, value_layout="bhsd", algo="direct"),
        "window bshd": lambda: ops.ms_deform_attn_forward(value, shapes, start, loc, attn, algo="window"),
        "window bhsd": lambda: ops.ms_deform_attn_forward(vh, shapes, start, loc, attn, value_layout="bhsd", algo="window"),
    }
    outs = {k: f().float() for k, f in arms.items()}
    ref = outs["direct bshd"]
    for k, o in outs.items():
        print(f"max |{k} - direct bshd| = {(o - ref).abs().max().item():.4g}")
    times = {k: [] for k in arms}
    for _ in range(rounds):
        for k, f in arms.items():
            times[k].append(timed(f, reps))
    for k, ts in times.items():
        t = min(ts)
        print(f"{k:12s}: min {t:7.1f} us  median {sorted(ts)[len(ts) // 2]:7.1f} us  -> {alg / t / 1e3:6.0f} GB/s = {alg / t / 1e3 / 8000 * 100:5.1f} % of 8 TB/s")
    t = timed(lambda: ops.value_to_head_major(value.view(B, S, 256)), reps)
    print(f"value_to_head_major: {t:.1f} us")


if __name__ == "__main__":
    main()
