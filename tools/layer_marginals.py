#!/usr/bin/env python3
"""Wall time per replayed step (B = 4, bf16, 2 image groups) as a function of the number of encoder / decoder layers: the
marginal cost of a layer in the two-group replay, to be compared with the sum of its kernels' durations."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from relation_detr_amd.graph import GraphedCall, ImageGroups  # noqa: E402
from relation_detr_amd.transformer import build_relation_transformer, select_detections  # noqa: E402


def main():
    dev = torch.device("cuda", 0)
    B, dtype = 4, torch.bfloat16
    feats, masks, pos = bench.build_pyramid(B, dev, seed=1000, dtype=dtype)
    sizes = torch.tensor([[800, 1333]] * B, device=dev)
    L = len(feats)
    flat = [*feats, *masks, *pos, sizes]
    for enc, dec in ((6, 6), (6, 3), (6, 1), (3, 6), (1, 6), (6, 6)):
        torch.manual_seed(0)
        net = build_relation_transformer(num_classes=91, d_ffn=2048, enc_layers=enc, dec_layers=dec, num_queries=900).eval().to(dev).to(dtype)

        @torch.no_grad()
        def forward_images(*t):
            classes, coords = net(list(t[:L]), list(t[L:2 * L]), list(t[2 * L:3 * L]))[:2]
            return select_detections(classes[-1].float(), coords[-1].float(), t[3 * L])

        run = GraphedCall(ImageGroups(forward_images, 2, device=dev), flat)
        for _ in range(5):
            run(*flat)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(30):
            run(*flat)
        torch.cuda.synchronize()
        el = (time.perf_counter() - t0) / 30
        print(f"enc {enc} dec {dec}: {el * 1e3:6.3f} ms/step", flush=True)
        del run, net


if __name__ == "__main__":
    main()
