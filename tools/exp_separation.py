"""How far apart are the bf16 and the fp32 stack's DETECTIONS as a function of the class-head weight scale?

VERDICT r02 weak #1 / next 1c: with the reference's init every class logit sits within ~1e-2 of the prior bias -- below
bf16 resolution (ulp 0.03 at |x| ~ 4.6) -- so `matched_frac` of the detections says nothing.  This tool measures, per
`class_scale` (bench.build_network): the share of two-stage proposals both routes pick, and the detection match at IoU
0.5 / 0.9, bf16 (two image groups, eager and replayed) against fp32, on bench.py's own images.

    python tools/exp_separation.py [scales...]        (default 1 8 32 128)
"""
import os
import sys

import torch

sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench  # noqa: E402
from relation_detr_amd.graph import GraphedCall, ImageGroups  # noqa: E402
from relation_detr_amd.transformer import select_detections  # noqa: E402

DEV = "cuda:0"


def make(net, L):
    @torch.no_grad()
    def fwd(*t):
        out = net(list(t[:L]), list(t[L:2 * L]), list(t[2 * L:3 * L]))
        return select_detections(out[0][-1].float(), out[1][-1].float(), t[3 * L]), out[3].float()
    return fwd


def main():
    scales = [float(v) for v in sys.argv[1:] if not v.startswith("--")] or [1.0, 8.0, 32.0, 128.0]
    exch = "--exchangeable" in sys.argv
    box_std = 0.0 if "--static-boxes" in sys.argv else 0.01
    B, L = 4, 4
    feats, masks, pos = bench.build_pyramid(B, DEV, seed=1000, dtype=torch.float32)
    sizes = torch.tensor([[800, 1333]] * B, device=DEV)
    in32 = [*feats, *masks, *pos, sizes]
    in16 = [t.to(torch.bfloat16) if t.is_floating_point() else t for t in in32]
    for sc in scales:
        net32 = bench.build_network(900, 0, class_scale=sc, exchangeable_queries=exch, box_head_std=box_std).to(DEV)
        net16 = bench.build_network(900, 0, class_scale=sc, exchangeable_queries=exch, box_head_std=box_std).to(DEV).to(torch.bfloat16)
        det32, prop32 = make(net32, L)(*in32)
        det32, prop32 = det32.clone(), prop32.clone()
        det16, prop16 = ImageGroups(make(net16, L), 2, device=DEV)(*in16)
        det16, prop16 = det16.clone(), prop16.clone()
        run = GraphedCall(ImageGroups(make(net16, L), 2, device=DEV), in16)
        rdet16 = run(*in16)[0].clone()
        torch.cuda.synchronize()
        # proposals: share of fp32's 900 boxes that bf16 also picked (any slot) / in the SAME slot
        d = torch.cdist(prop16.double(), prop32.double(), p=float("inf"))            # [B, 900, 900]
        any_slot = (d.min(1)[0] < 2e-2).float().mean().item()
        same_slot = ((prop16 - prop32).abs().max(-1)[0] < 2e-2).float().mean().item()
        logits = net32.encoder_class_head.weight.std().item()
        out = {"class_scale": sc, "exchangeable_queries": exch, "box_head_std": box_std, "class_head_weight_std": round(logits, 4), "proposals_common": round(any_slot, 4),
               "proposals_same_slot": round(same_slot, 4)}
        for thr in (0.5, 0.9):
            m = bench.detection_drift(det16, det32, iou_thr=thr)
            out[f"bf16_vs_fp32@{thr}"] = {k: (round(v, 4) if isinstance(v, float) else v) for k, v in m.items()}
        out["replay_vs_eager@0.9"] = round(bench.detection_drift(rdet16, det16, iou_thr=0.9)["matched_frac"], 4)
        out["replay_bit_identical"] = bool(torch.equal(rdet16, det16))
        s32 = det32[..., 4]
        out["fp32_score_range"] = [round(s32.min().item(), 4), round(s32.max().item(), 4)]
        out["fp32_distinct_scores"] = int(torch.unique(s32).numel())
        out["fp32_distinct_boxes"] = int(torch.unique(det32[..., :4].reshape(-1, 4), dim=0).shape[0])
        print(out, flush=True)
        del net32, net16, run


if __name__ == "__main__":
    main()
