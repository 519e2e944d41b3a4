import os, sys, torch
from torch.profiler import ProfilerActivity, profile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import bench
dev = torch.device("cuda", 0)
net = bench.build_network(900, 0).to(dev).to(torch.bfloat16)
feats, masks, pos = bench.build_pyramid(4, dev, seed=1000, dtype=torch.bfloat16)
@torch.no_grad()
def step():
    return net(feats, masks, pos)
for _ in range(2): step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True, with_stack=True) as prof:
    step(); torch.cuda.synchronize()
for ev in prof.key_averages(group_by_input_shape=True, group_by_stack_n=6):
    if ev.key in ("aten::add", "aten::mul", "aten::copy_", "aten::clone", "aten::cat", "aten::sub", "aten::div") and ev.device_time_total > 150:
        print(f"{ev.key:12s} n={ev.count:3d} cuda={ev.device_time_total:9.1f}us shapes={ev.input_shapes}")
        for s in ev.stack[:6]:
            if "relation_detr_amd" in s or "bench" in s: print("      ", s)
