#!/usr/bin/env python3
"""Development check of the sweep kernel (csrc/msda_sweep.hip): parity against the direct kernel on a few pyramids, then the
duration at the encoder shape of BASELINE.json configs[1] next to the direct kernel.
    python3 tools/sweep_check.py [reps] [B]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import bench  # noqa: E402
from relation_detr_amd import ops  # noqa: E402

DEV = torch.device("cuda", 0)
ALGO = os.environ.get("ALGO", "sweep")


def timed(fn, reps, warm=20):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


def parity():
    import test_gpu_window as tw
    cases = [(tw.R50, 1, 3.0, 0.0, False), (tw.R50, 2, 4.0, 0.0, True), (tw.R50, 1, 30.0, 0.0, False),
             ([(64, 96), (32, 48), (16, 24), (8, 12)], 2, 4.0, 0.05, False),
             ([(75, 61), (38, 31), (19, 16), (10, 8)], 3, 6.0, 0.0, True),
             ([(70, 70), (35, 35), (18, 18), (9, 9)], 1, 2.0, 1.0, False),
             ([(160, 24), (80, 12), (40, 6), (20, 3)], 2, 3.0, 0.02, True),
             ([(12, 20), (6, 10), (3, 5), (2, 3)], 2, 2.0, 0.1, True)]
    ok = True
    for shapes, B, spread, scatter, poison in cases:
        value, shp, start, loc, attn, S, L = tw._encoder_inputs(shapes, B, spread, seed=int(spread * 7) + B, scatter=scatter, poison=poison)
        rest = (shp.to(DEV), start.to(DEV), loc.to(DEV), attn.to(DEV))
        v = value.to(DEV)
        direct = ops.ms_deform_attn_forward(v, *rest, algo="direct").float()
        for layout, vv in (("bshd", v), ("bhsd", tw._head_major(v))):
            out = ops.ms_deform_attn_forward(vv, *rest, value_layout=layout, algo=ALGO).float()
            torch.cuda.synchronize()
            err = (out - direct).abs()
            bad = err > 2.0 ** -7 * direct.abs() + 1e-3
            nb = int(bad.sum())
            print(f"{shapes[0]} B={B} spread={spread} scatter={scatter} {layout}: max err {err.max().item():.3g}, bad {nb}, finite {bool(torch.isfinite(out).all())}")
            if nb:
                ok = False
                idx = bad.nonzero()[:6].tolist()
                print("   first bad (b, q, ch):", idx)
    return ok


def main():
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 50
    B = int(sys.argv[2]) if len(sys.argv) > 2 else 4
    ok = parity()
    print("PARITY", "OK" if ok else "FAILED")
    value, shapes, start, loc, attn, S, L = bench.encoder_kernel_inputs(B, DEV, torch.bfloat16)
    vh = value.permute(0, 2, 1, 3).contiguous()
    alg = bench.msda_algorithmic_bytes(B, S, S, L, 4, 8, 32, 2)
    arms = {
        "direct bhsd": lambda: ops.ms_deform_attn_forward(vh, shapes, start, loc, attn, value_layout="bhsd", algo="direct"),
        "window bhsd": lambda: ops.ms_deform_attn_forward(vh, shapes, start, loc, attn, value_layout="bhsd", algo="window"),
        "sweep  bhsd": lambda: ops.ms_deform_attn_forward(vh, shapes, start, loc, attn, value_layout="bhsd", algo="sweep"),
        "sweep  bshd": lambda: ops.ms_deform_attn_forward(value, shapes, start, loc, attn, algo="sweep"),
    }
    outs = {k: f().float() for k, f in arms.items()}
    ref = outs["direct bhsd"]
    for k, o in outs.items():
        print(f"max |{k} - direct| = {(o - ref).abs().max().item():.4g}")
    for rnd in range(3):
        for k, f in arms.items():
            t = timed(f, reps, warm=100 if rnd == 0 else 20)
            print(f"round {rnd} {k}: {t:7.1f} us -> {alg / t / 1e3:6.0f} GB/s = {alg / t / 1e3 / 8000 * 100:5.1f} % of 8 TB/s", flush=True)


if __name__ == "__main__":
    main()
