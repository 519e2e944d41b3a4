import os, sys
os.environ["RDETR_BENCH_TUNABLEOP"] = "0"
pass
pass
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
import bench
DEV = "cuda:0"
net = bench.build_network(900, 0).to(DEV)
feats, masks, pos = bench.build_pyramid(1, "cpu", 7)
masks[0][0, :, 150:] = True
for l in range(1, 4):
    masks[l][0, :, masks[l].shape[2] * 150 // 168:] = True
print("start", flush=True)
with torch.no_grad():
    out = net([f.to(DEV) for f in feats], [m.to(DEV) for m in masks], [p.to(DEV) for p in pos])
torch.cuda.synchronize()
print("ok", [o.shape for o in out], flush=True)
