"""How well do independent image-group chains overlap?  G graphs of b images each, free-running on G streams, against ONE of them
running alone.  If the period of a chain grows with G although the chip is far from full (b = 1), the launch path -- not the
kernels -- is what the groups share.

    python tools/exp_chain_overlap.py
"""
import os
import sys
import time

import torch

sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench  # noqa: E402
from relation_detr_amd.graph import GraphedCall  # noqa: E402
from relation_detr_amd.transformer import select_detections  # noqa: E402

dev = torch.device("cuda", 0)
L = 4
net = bench.build_network(900, 0).to(dev).to(torch.bfloat16)


@torch.no_grad()
def fwd(*t):
    c, b = net(list(t[:L]), list(t[L:2 * L]), list(t[2 * L:3 * L]))[:2]
    return select_detections(c[-1].float(), b[-1].float(), t[3 * L])


def inputs(b, seed):
    feats, masks, pos = bench.build_pyramid(b, dev, seed=seed, dtype=torch.bfloat16)
    return [*feats, *masks, *pos, torch.tensor([[800, 1333]] * b, device=dev)]


def run(b, G, steps=40, warm=10):
    ins = [inputs(b, 1000 + g) for g in range(G)]
    graphs = [GraphedCall(fwd, i) for i in ins]
    streams = [torch.cuda.Stream(device=dev) for _ in range(G)]
    for s in streams:
        s.wait_stream(torch.cuda.current_stream())

    def step():
        for g, i, s in zip(graphs, ins, streams):
            with torch.cuda.stream(s):
                g(*i)
    for _ in range(warm):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    torch.cuda.synchronize()
    t = (time.perf_counter() - t0) / steps
    print(f"{G} chain(s) of {b} image(s): period {t * 1e3:6.3f} ms   {G * b / t:7.1f} images/s", flush=True)
    del graphs
    torch.cuda.synchronize()


for b, G in ((1, 1), (1, 2), (1, 4), (2, 1), (2, 2), (2, 3), (4, 1), (4, 2)):
    run(b, G)
