"""rdetr_encoder_proj_k256_bf16 (csrc/proj.hip) against the two launches it replaces (value_proj_head_major + the N = 384 library
GEMM), each replayed as a HIP graph of 20 calls: us per call at one image group's rows (44,646) and at 89,292."""
import os
import sys

import torch

sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from relation_detr_amd import ops  # noqa: E402
from tools.time_qpos import timed  # noqa: E402

dev = "cuda:0"
g = torch.Generator().manual_seed(0)
wv = (torch.randn(256, 256, generator=g) * 0.06).to(torch.bfloat16).to(dev)
bv = (torch.randn(256, generator=g) * 0.1).to(torch.bfloat16).to(dev)
wq = (torch.randn(384, 256, generator=g) * 0.06).to(torch.bfloat16).to(dev)
bq = (torch.randn(384, generator=g) * 0.1).to(torch.bfloat16).to(dev)
for B in (2, 4):
    S = 22323
    x = torch.randn(B, S, 256, generator=g).to(torch.bfloat16).to(dev)
    xq = torch.randn(B, S, 256, generator=g).to(torch.bfloat16).to(dev)
    mask = torch.zeros(B, S, dtype=torch.bool, device=dev)
    fused = timed(lambda: ops.encoder_proj(x, xq, wv, bv, wq, bq, mask))
    sep = timed(lambda: (ops.value_proj_head_major(x, wv, bv, mask), torch.nn.functional.linear(xq, wq, bq)))
    print(f"rows {B * S}: one kernel {fused:.1f} us | value_proj_head_major + N=384 library GEMM {sep:.1f} us")
