#!/usr/bin/env python3
"""Run only the encoder-shape MSDA gather kernel (BASELINE.json configs[1]: B=4, S=Nq=22,323, 4 levels)
so that rocprofv3 traces / PMC passes of it stay small.

    rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/x -- python3 tools/profile_msda.py [fp32|bf16] [reps]
    rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/y -- python3 tools/profile_msda.py fp32 5
"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
    dtype = torch.bfloat16 if (len(sys.argv) > 1 and sys.argv[1] == "bf16") else torch.float32
    reps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
    B = int(sys.argv[3]) if len(sys.argv) > 3 else 4
    dev = torch.device("cuda", 0)
    t, S, L = bench.time_encoder_kernel(B, dev, dtype, reps=reps)
    alg = bench.msda_algorithmic_bytes(B, S, S, L, 4, 8, 32, 2 if dtype == torch.bfloat16 else 4)
    print(f"{dtype} B={B}: {t*1e6:.1f} us/launch, algorithmic {alg/1e6:.1f} MB -> {alg/t/1e9:.0f} GB/s ({alg/t/8e12*100:.1f}% of 8 TB/s)")
    # decoder shape: Nq = 900 box queries per image (centre +- U(-.5,.5)*wh)
    import relation_detr_amd as rd
    g = torch.Generator().manual_seed(5)
    Nq = 900
    value, shapes, start, _, _, _, _ = bench.encoder_kernel_inputs(B, dev, dtype)
    cxcy = torch.rand(B, Nq, 1, 1, 1, 2, generator=g) * 0.8 + 0.1
    wh = torch.rand(B, Nq, 1, 1, 1, 2, generator=g) * 0.48 + 0.02
    loc = (cxcy + (torch.rand(B, Nq, 8, L, 4, 2, generator=g) - 0.5) * wh).contiguous().to(dev)
    attn = torch.softmax(torch.randn(B, Nq, 8, L * 4, generator=g), -1).view(B, Nq, 8, L, 4).contiguous().to(dev)
    for _ in range(3):
        rd.ms_deform_attn_forward(value, shapes, start, loc, attn, 64)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        rd.ms_deform_attn_forward(value, shapes, start, loc, attn, 64)
    e1.record()
    torch.cuda.synchronize()
    td = e0.elapsed_time(e1) / reps * 1e-3
    algd = bench.msda_algorithmic_bytes(B, S, Nq, L, 4, 8, 32, 2 if dtype == torch.bfloat16 else 4)
    print(f"   decoder Nq=900: {td*1e6:.1f} us/launch, algorithmic {algd/1e6:.1f} MB -> {algd/td/1e9:.0f} GB/s")


if __name__ == "__main__":
    main()
