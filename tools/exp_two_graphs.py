"""Two image groups as TWO graphs on two streams, free-running (no join between steps), against bench.py's one graph with a
fork / join per step.  Round 2 tried this and hit a GPU memory fault while the second graph was WARMED UP during replays of
the first; GraphedCall now builds with the device idle (graph.py), so both graphs are built first, then replayed.

    python tools/exp_two_graphs.py [steps]
"""
import os
import sys
import time

import torch

sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench  # noqa: E402
from relation_detr_amd.graph import GraphedCall, ImageGroups  # noqa: E402
from relation_detr_amd.transformer import select_detections  # noqa: E402

dev = torch.device("cuda", 0)
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 60
B, L = 4, 4
feats, masks, pos = bench.build_pyramid(B, dev, seed=1000, dtype=torch.bfloat16)
sizes = torch.tensor([[800, 1333]] * B, device=dev)
flat = [*feats, *masks, *pos, sizes]
net = bench.build_network(900, 0).to(dev).to(torch.bfloat16)


@torch.no_grad()
def fwd(*t):
    c, b = net(list(t[:L]), list(t[L:2 * L]), list(t[2 * L:3 * L]))[:2]
    return select_detections(c[-1].float(), b[-1].float(), t[3 * L])


def timed(step, n, warm=15):
    for _ in range(warm):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        step()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n


# (a) bench.py's launch: one graph, two groups forked and joined inside it
one = GraphedCall(ImageGroups(fwd, 2, device=dev), flat)
ta = timed(lambda: one(*flat), steps)
ref = one(*flat).clone()
torch.cuda.synchronize()
print(f"(a) one graph, fork/join per step:            {ta * 1e3:.3f} ms/step  {B / ta:7.1f} images/s", flush=True)

# (b, c) one graph per group, each on its own stream, no join between steps
halves = [[t[:2].contiguous() for t in flat], [t[2:].contiguous() for t in flat]]
graphs = [GraphedCall(fwd, h) for h in halves]          # both built with the device idle, before any replay
for prios, tag, offset_ms in (((0, 0), "equal priority", 0.0), ((0, 0), "group 1 starts 1 ms late", 1.0), ((0, 0), "group 1 starts 2 ms late", 2.0),
                              ((0, 0), "group 1 starts 3 ms late", 3.0), ((-1, 0), "group 0 high priority", 0.0)):
    streams = [torch.cuda.Stream(device=dev, priority=p) for p in prios]
    for s in streams:
        s.wait_stream(torch.cuda.current_stream())
    if offset_ms:
        with torch.cuda.stream(streams[1]):
            torch.cuda._sleep(int(offset_ms * 1e-3 * 2.1e9))     # the phase offset persists: nothing joins the two streams

    def step():
        for g, h, s in zip(graphs, halves, streams):
            with torch.cuda.stream(s):
                g(*h)

    tb = timed(step, steps, warm=15 if not offset_ms else 0)   # (a warm-up loop ends in a synchronize: it would undo the offset)
    torch.cuda.synchronize()
    out = torch.cat([g(*h) for g, h in zip(graphs, halves)], 0)
    torch.cuda.synchronize()
    same = torch.equal(out, ref)
    print(f"(b) two graphs, two streams, {tag:26s}: {tb * 1e3:.3f} ms/step  {B / tb:7.1f} images/s   detections equal to (a): {same}", flush=True)
