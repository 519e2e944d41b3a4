"""Does the dominant kernel's measured duration depend on how long it has been running?  bench.py times it with 3 warm-up
and 20 timed launches right after an idle phase (inputs are built on the host); the same-process A/B tools (tools/ab_msda.py)
run hundreds of launches and see 105-108 us where bench.py reports 118-122 us.  This prints the average over windows of
20 launches, back to back, for 40 windows -- if the figure falls with time it is clock ramp-up, not the kernel."""
import os
import sys

import torch

sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench  # noqa: E402
import relation_detr_amd as rd  # noqa: E402

dev = torch.device("cuda", 0)
B = 4
value, shapes, start, loc, attn, S, L = bench.encoder_kernel_inputs(B, dev, torch.bfloat16)
value = value.permute(0, 2, 1, 3).contiguous()
torch.cuda.synchronize()
out = []
for w in range(40):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        rd.ms_deform_attn_forward(value, shapes, start, loc, attn, 64, value_layout="bhsd")
    e1.record()
    torch.cuda.synchronize()
    out.append(e0.elapsed_time(e1) / 20 * 1e3)
print("us per launch, windows of 20 launches:", " ".join(f"{v:.1f}" for v in out))
# the same with a host-side pause between windows (the GPU idles ~50 ms)
import time
out2 = []
for w in range(10):
    time.sleep(0.05)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        rd.ms_deform_attn_forward(value, shapes, start, loc, attn, 64, value_layout="bhsd")
    e1.record()
    torch.cuda.synchronize()
    out2.append(e0.elapsed_time(e1) / 20 * 1e3)
print("after a 50-ms idle gap each:", " ".join(f"{v:.1f}" for v in out2))
