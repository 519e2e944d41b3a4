"""Steady-state duration of the benched gather kernel (bf16, head-major, B = 4 encoder shape; operator form): 300 warm-up
launches (clock ramp, tools/exp_kernel_timing.py), then the average of 200, three times.  RDETR_LIB_PATH selects a build."""
import os
import sys

import torch

sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench  # noqa: E402
import relation_detr_amd as rd  # noqa: E402

dev = torch.device("cuda", 0)
layout = sys.argv[1] if len(sys.argv) > 1 else "bhsd"
value, shapes, start, loc, attn, S, L = bench.encoder_kernel_inputs(4, dev, torch.bfloat16)
if layout == "bhsd":
    value = value.permute(0, 2, 1, 3).contiguous()
run = lambda: rd.ms_deform_attn_forward(value, shapes, start, loc, attn, 64, value_layout=layout, algo="direct")
for _ in range(300):
    run()
res = []
for _ in range(3):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(200):
        run()
    e1.record()
    torch.cuda.synchronize()
    res.append(e0.elapsed_time(e1) / 200 * 1e3)
print(os.environ.get("RDETR_LIB_PATH", "product build"), layout, "us per launch:", " ".join(f"{v:.1f}" for v in res))
