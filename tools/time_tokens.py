import os, sys, torch
sys.path.insert(0, "/root/repo" if os.path.isdir("/root/repo") else os.getcwd())
from relation_detr_amd import ops
from tools.time_linear import timed
for B in (2, 4):
    lv = [torch.randn(B, 256, h, w, device="cuda").bfloat16() for h, w in ((100, 168), (50, 84), (25, 42), (13, 21))]
    em = list(torch.randn(4, 256, device="cuda").bfloat16())
    t = timed(lambda: ops.tokens_from_levels(lv, add_vecs=em))
    nbytes = 2 * sum(x.numel() for x in lv) * 2
    print(f"B={B}: tokens_from_levels {t*1e6:6.1f} us  ({nbytes/t/1e12:.2f} TB/s)", flush=True)
