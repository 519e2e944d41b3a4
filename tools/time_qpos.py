"""rdetr_query_pos_k256_bf16 (csrc/qpos.hip) against the unfused sequence it replaces (4 library GEMMs + scaled_pos), each
replayed as a HIP graph of 20 back-to-back calls: us per call at 1,800 and 3,600 rows."""
import os
import sys

import torch

sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from relation_detr_amd import ops  # noqa: E402
from relation_detr_amd.transformer import MLP  # noqa: E402

dev = "cuda:0"
torch.manual_seed(0)
head = MLP(512, 256, 256, 2).to(dev).to(torch.bfloat16)
scale = MLP(256, 256, 256, 2).to(dev).to(torch.bfloat16)


def timed(fn, reps=20, rounds=30):
    with torch.no_grad():
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            for _ in range(reps):
                fn()
        for _ in range(5):
            g.replay()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(rounds):
            g.replay()
        e1.record()
        torch.cuda.synchronize()
    return e0.elapsed_time(e1) / (reps * rounds) * 1e3


if __name__ == "__main__":
  for rows in (1800, 3600):
      emb = torch.randn(2, rows // 2, 512, device=dev).to(torch.bfloat16)
      q = torch.randn(2, rows // 2, 256, device=dev).to(torch.bfloat16)
      fused = timed(lambda: ops.query_pos_k256(emb, q, head.layers, scale.layers))
      fused0 = timed(lambda: ops.query_pos_k256(emb, q, head.layers, None))
      unfused = timed(lambda: ops.scaled_pos(head(emb), scale(q), q))
      unfused0 = timed(lambda: head(emb))
      print(f"rows {rows}: fused {fused:.1f} us (layer 0: {fused0:.1f}) | unfused 4 GEMMs + scaled_pos {unfused:.1f} us (layer 0, 2 GEMMs: {unfused0:.1f})")
