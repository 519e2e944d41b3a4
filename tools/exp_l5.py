#!/usr/bin/env python3
"""Where the 5-level (FocalNet) gather loses against the 4-level one: the same kernel timed on four level tables, bf16,
head-major value, operator form (materialised locations / weights), steady clocks (untimed launches first).
    python3 tools/exp_l5.py"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import relation_detr_amd as rd  # noqa: E402

R50 = bench.CONFIGS["r50"]["shapes"] if "shapes" in bench.CONFIGS["r50"] else bench.R50_SHAPES
FOC = bench.CONFIGS["focalnet"]["shapes"]


def run(name, shapes, B, reps=30):
    dev = torch.device("cuda", 0)
    value, sh, start, loc, attn, S, L = bench.encoder_kernel_inputs(B, dev, torch.bfloat16, shapes)
    value = value.permute(0, 2, 1, 3).contiguous()
    for _ in range(3 * reps):
        rd.ms_deform_attn_forward(value, sh, start, loc, attn, 64, value_layout="bhsd")
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        rd.ms_deform_attn_forward(value, sh, start, loc, attn, 64, value_layout="bhsd")
    e1.record()
    torch.cuda.synchronize()
    t = e0.elapsed_time(e1) / reps * 1e-3
    samples = B * S * 8 * L * 4
    alg = bench.msda_algorithmic_bytes(B, S, S, L, 4, 8, 32, 2)
    print(f"{name:44s} B={B} S={S:7d} L={L}: {t*1e6:8.1f} us  {t/samples*1e12:6.2f} ps/sample  frac {alg/t/8e12:.3f}", flush=True)
    del value, loc, attn
    torch.cuda.empty_cache()


def run_fused(name, shapes, B, reps=30):
    """The module path's form: raw offsets / logits as column slices of ONE packed projection output [rows, 3*H*L*P] bf16."""
    from relation_detr_amd import ops
    dev = torch.device("cuda", 0)
    value, sh, start, loc, attn, S, L = bench.encoder_kernel_inputs(B, dev, torch.bfloat16, shapes)
    del loc, attn
    value = value.permute(0, 2, 1, 3).contiguous()
    g = torch.Generator().manual_seed(7)
    n_off, n_lg = 8 * L * 4 * 2, 8 * L * 4
    packed = torch.randn(B, S, n_off + n_lg, generator=g).to(dev).to(torch.bfloat16)
    off = packed[..., :n_off].view(B, S, 8, L, 4, 2)
    lg = packed[..., n_off:].view(B, S, 8, L * 4)
    refs = []
    for h, w in shapes:
        ys, xs = torch.meshgrid((torch.arange(h) + 0.5) / h, (torch.arange(w) + 0.5) / w, indexing="ij")
        refs.append(torch.stack([xs.reshape(-1), ys.reshape(-1)], -1))
    ref = torch.cat(refs, 0)[None, :, None, :].expand(B, S, L, 2).contiguous().to(dev)
    call = lambda: ops.ms_deform_attn_forward_fused(value, sh, start, off, lg, ref, value_layout="bhsd")
    for _ in range(3 * reps):
        call()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        call()
    e1.record()
    torch.cuda.synchronize()
    t = e0.elapsed_time(e1) / reps * 1e-3
    samples = B * S * 8 * L * 4
    print(f"fused  {name:37s} B={B} S={S:7d} L={L}: {t*1e6:8.1f} us  {t/samples*1e12:6.2f} ps/sample", flush=True)
    del value, packed
    torch.cuda.empty_cache()


def main():
    print("library:", os.environ.get("RDETR_LIB_PATH", "product"), flush=True)
    run_fused("r50 4 levels", list(R50), 4)
    run_fused("focalnet 5 levels", list(FOC), 2)
    run_fused("focalnet, its first 4 levels", list(FOC)[:4], 2)
    run("r50 4 levels", list(R50), 4)
    run("r50 + a fifth level (7,11)", list(R50) + [(7, 11)], 4)
    run("focalnet, its first 4 levels", list(FOC)[:4], 2)
    run("focalnet 5 levels", list(FOC), 2)
    run("focalnet levels 1..4 (4 levels, S = 51k)", list(FOC)[1:], 4)


if __name__ == "__main__":
    main()
