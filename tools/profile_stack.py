#!/usr/bin/env python3
"""torch.profiler view of one bench step (bf16, B=4): operator-level GPU time with input shapes, to see which glue
operators around the hot-path kernels dominate.     python3 tools/profile_stack.py [bf16|fp32]"""
import os
import sys

import torch
from torch.profiler import ProfilerActivity, profile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from relation_detr_amd.transformer import select_detections  # noqa: E402


def main():
    dtype = torch.float32 if (len(sys.argv) > 1 and sys.argv[1] == "fp32") else torch.bfloat16
    dev = torch.device("cuda", 0)
    net = bench.build_network(900, 0).to(dev).to(dtype)
    feats, masks, pos = bench.build_pyramid(4, dev, seed=1000, dtype=dtype)
    sizes = torch.tensor([[800, 1333]] * 4, device=dev)

    @torch.no_grad()
    def step():
        classes, coords = net(feats, masks, pos)[:2]
        return select_detections(classes[-1].float(), coords[-1].float(), sizes)

    for _ in range(3):
        step()
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
        for _ in range(3):
            step()
        torch.cuda.synchronize()
    print(prof.key_averages(group_by_input_shape=True).table(sort_by="cuda_time_total", row_limit=45, max_name_column_width=48,
                                                              max_shapes_column_width=70))
    # the small torch operators (what is left of the launch chains): calls per step by name and shapes
    rows = [e for e in prof.key_averages(group_by_input_shape=True) if e.key.startswith("aten::") and e.device_time_total > 0]
    rows.sort(key=lambda e: -e.count)
    print("\naten operators with device time, by call count (3 steps):")
    for e in rows[:60]:
        print(f"  {e.count:5d} x {e.key:32s} {e.device_time_total / max(e.count, 1):8.1f} us each   {str(e.input_shapes)[:110]}")


if __name__ == "__main__":
    main()
