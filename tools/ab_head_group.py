#!/usr/bin/env python3
"""Same-box A/B of the direct MSDA kernel's HEAD GROUP size (development library: rdetr_dev_set_msda_head_group_log2): G = 1, 2, 4,
8 heads taking turns over the same query tile inside one XCD.  For every shape and form: bit-identity of the output with G = 1,
sustained-clock time per launch.
    RDETR_LIB_PATH=relation_detr_amd/librelation_detr_amd_dev.so python3 tools/ab_head_group.py"""
import ctypes
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import relation_detr_amd as rd  # noqa: E402
from relation_detr_amd import _lib  # noqa: E402


def fused_inputs(B, dev, level_shapes):
    value, shapes, start, loc, attn, S, L = bench.encoder_kernel_inputs(B, dev, torch.bfloat16, level_shapes)
    vh = value.permute(0, 2, 1, 3).contiguous()
    g = torch.Generator().manual_seed(321)
    k = torch.arange(1, 5, dtype=torch.float32).view(1, 1, 1, 1, 4, 1)
    off = (torch.randn(B, S, 8, L, 4, 2, generator=g) * k).to(torch.bfloat16).to(dev)
    logits = torch.randn(B, S, 8, L * 4, generator=g).to(torch.bfloat16).to(dev)
    refs = []
    for h, w in level_shapes:
        ys, xs = torch.meshgrid((torch.arange(h) + 0.5) / h, (torch.arange(w) + 0.5) / w, indexing="ij")
        refs.append(torch.stack([xs.reshape(-1), ys.reshape(-1)], -1))
    ref = torch.cat(refs, 0)[None, :, None, :].expand(B, S, L, 2).contiguous().to(dev)
    return vh, shapes, start, off, logits, ref


def main():
    dev = torch.device("cuda", 0)
    lib = ctypes.CDLL(_lib.LIB_PATH)
    reps = int(os.environ.get("REPS", "40"))
    for name, B in (("r50", 4), ("r50", 2), ("focalnet", 2)):
        cfg = bench.CONFIGS[name]
        value, shapes, start, loc, attn, S, L = bench.encoder_kernel_inputs(B, dev, torch.bfloat16, cfg["shapes"])
        vh = value.permute(0, 2, 1, 3).contiguous()
        forms = {
            "operator, head-major value": lambda: rd.ops.ms_deform_attn_forward(vh, shapes, start, loc, attn, value_layout="bhsd", algo="direct"),
            "operator, [B,S,H,D] value": lambda: rd.ops.ms_deform_attn_forward(value, shapes, start, loc, attn, value_layout="bshd", algo="direct"),
        }
        fi = fused_inputs(B, dev, cfg["shapes"])
        forms["fused producer, head-major value"] = lambda: rd.ops.ms_deform_attn_forward_fused(fi[0], fi[1], fi[2], fi[3], fi[4], fi[5], value_layout="bhsd")
        for form, fn in forms.items():
            base = None
            for hg in (0, 1, 2, 3, 0):
                lib.rdetr_dev_set_msda_head_group_log2(hg)
                out = fn().clone()
                torch.cuda.synchronize()
                if base is None:
                    base = out
                same = torch.equal(out, base)
                for _ in range(2 * reps):
                    fn()
                t = bench._timed_launches(fn, reps)
                print(f"{name} B={B}  {form:36s} G={1 << hg}  {t * 1e6:8.1f} us  bit-identical to G=1: {same}", flush=True)
        del value, vh, loc, attn, fi
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
