#!/usr/bin/env python3
"""Resident-levels MSDA kernel (csrc/msda_res.hip) against the query-run kernel on the same inputs: bit-identity (operator form),
closeness (fused form: the softmax normaliser is summed in another order), time per launch at sustained clocks.
    python3 tools/res_check.py [r50|focalnet ...]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import relation_detr_amd as rd  # noqa: E402
from tools.ab_head_group import fused_inputs  # noqa: E402


def main():
    dev = torch.device("cuda", 0)
    import ctypes
    from relation_detr_amd import _lib
    devlib = ctypes.CDLL(_lib.LIB_PATH)
    reps = int(os.environ.get("REPS", "40"))
    cases = [(n, int(b)) for n, b in (c.split(":") for c in sys.argv[1:])] or [("r50", 4), ("r50", 2), ("r50", 1), ("r50", 3), ("focalnet", 2)]
    for name, B in cases:
        cfg = bench.CONFIGS[name]
        value, shapes, start, loc, attn, S, L = bench.encoder_kernel_inputs(B, dev, torch.bfloat16, cfg["shapes"])
        vh = value.permute(0, 2, 1, 3).contiguous()
        fi = fused_inputs(B, dev, cfg["shapes"])
        forms = {
            "operator": lambda algo: rd.ops.ms_deform_attn_forward(vh, shapes, start, loc, attn, value_layout="bhsd", algo=algo),
            "fused": lambda algo: rd.ops.ms_deform_attn_forward_fused(fi[0], fi[1], fi[2], fi[3], fi[4], fi[5], value_layout="bhsd", algo=algo),
        }
        for form, fn in forms.items():
            ref = fn("direct").float()
            out = fn("resident").float()
            torch.cuda.synchronize()
            diff = (out - ref).abs()
            print(f"{name} B={B} {form:9s} max |resident - direct| = {diff.max().item():.3e}  (|direct| max {ref.abs().max().item():.3f}, "
                  f"differing elements {int((diff > 0).sum())} of {diff.numel()})  bit-identical: {torch.equal(out, ref)}", flush=True)
            variants = [("direct", 16), ("resident", 16), ("direct", 16), ("resident", 16)]
            if hasattr(devlib, "rdetr_dev_set_res_waves"):                                  # development library: waves per workgroup
                variants = [("direct", 12, 0), ("resident", 12, 0), ("resident", 12, 2), ("resident", 12, 1), ("direct", 12, 0), ("resident", 12, 0), ("resident", 12, 2), ("resident", 12, 1), ("resident", 12, 0), ("resident", 12, 2), ("resident", 12, 1)]

            for v in variants:
                algo, waves, sg = (v + (0,))[:3]
                if hasattr(devlib, "rdetr_dev_set_res_waves"):
                    devlib.rdetr_dev_set_res_waves(waves)
                if hasattr(devlib, "rdetr_dev_set_res_max_teams"):
                    devlib.rdetr_dev_set_res_max_teams(sg)
                f = lambda: fn(algo)
                same = torch.equal(f().float(), ref) if algo == "resident" else True
                for _ in range(2 * reps):
                    f()
                t = bench._timed_launches(f, reps)
                print(f"    {algo:9s} waves {waves:2d} planes in flight per XCD (0 = all) {sg} {t * 1e6:8.1f} us   bit-identical to direct: {same}", flush=True)
            if hasattr(devlib, "rdetr_dev_set_res_waves"):
                devlib.rdetr_dev_set_res_waves(12)
                if hasattr(devlib, "rdetr_dev_set_res_max_teams"):
                    devlib.rdetr_dev_set_res_max_teams(0)
        del value, vh, loc, attn, fi
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
