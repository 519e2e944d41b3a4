import sys, os, torch, numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo")); sys.path.insert(0, os.path.join(sys.path[0], "tests"))
import relation_detr_amd as rd
from helpers import pyramid
shapes=[(9, 14), (5, 7), (3, 4)]
shp,start,S=pyramid(shapes); L=3; B=1
g=torch.Generator().manual_seed(1)
value=torch.randn(B,S,8,32,generator=g)
refs=[]
for h,w in shapes:
    ys,xs=torch.meshgrid((torch.arange(h)+0.5)/h,(torch.arange(w)+0.5)/w,indexing="ij"); refs.append(torch.stack([xs.reshape(-1),ys.reshape(-1)],-1))
ref=torch.cat(refs,0)
wh=shp.flip(-1).float()
off=torch.randn(B,S,8,L,4,2,generator=g)*2.0/wh.view(1,1,1,L,1,2)
loc=(ref[None,:,None,None,None,:]+off).contiguous()
attn=torch.softmax(torch.randn(B,S,8,L*4,generator=g),-1).view(B,S,8,L,4)
args=(value.cuda(),shp.cuda(),start.cuda(),loc.cuda(),attn.cuda(),64)
out=rd.ms_deform_attn_forward(*args).cpu()
os.environ["RDETR_MSDA_ENCODER"]="qrun"
d=rd.ms_deform_attn_forward(*args).cpu()
err=(out-d).abs()
print("max err", err.max().item())
e=err.view(B,S,8,32)
print("per channel max err:", [round(x,3) for x in e.amax(dim=(0,1,2)).tolist()])
print("per head:", [round(x,3) for x in e.amax(dim=(0,1,3)).tolist()])
print("per query (first 30):", [round(x,2) for x in e.amax(dim=(0,2,3))[:30].tolist()])
print("ratio sample", (out.view(B,S,8,32)[0,5,0,:8]).tolist(), d.view(B,S,8,32)[0,5,0,:8].tolist())
