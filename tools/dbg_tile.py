import sys, torch, numpy as np
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from relation_detr_amd import ops
from helpers import pyramid
DEV='cuda:0'
shapes=[(64,96),(32,48),(16,24),(8,12)]
shp,start,S=pyramid(shapes); L=4; B=1
def refs():
    r=[]
    for h,w in shapes:
        ys,xs=torch.meshgrid((torch.arange(h)+0.5)/h,(torch.arange(w)+0.5)/w,indexing='ij'); r.append(torch.stack([xs.reshape(-1),ys.reshape(-1)],-1))
    return torch.cat(r,0)
loc=refs()[None,:,None,None,None,:].expand(B,S,8,L,4,2).contiguous()
for lv in range(4):
    value=torch.zeros(B,S,8,32)
    st=int(start[lv]); n=shapes[lv][0]*shapes[lv][1]
    value[:,st:st+n]= (torch.arange(32).float()+1).view(1,1,1,32)*(torch.arange(8).float()+1).view(1,1,8,1)
    attn=torch.zeros(B,S,8,L,4); attn[:,:,:,lv,:]=0.25
    args=(value.to(torch.bfloat16).to(DEV),shp.to(DEV),start.to(DEV),loc.to(DEV),attn.to(DEV))
    o=ops.ms_deform_attn_forward_strategy('tiled',*args).float().cpu()
    d=ops.ms_deform_attn_forward_strategy('direct',*args).float().cpu()
    err=(o-d).abs()
    print('level',lv,'max err',err.max().item(),'rows bad',int((err.amax(-1)>1e-2).sum()),'of',S)
    bad=(err.amax(-1)>1e-2)[0].nonzero().flatten()
    print('  first bad rows',bad[:20].tolist())
    print('  tiled row0 ch0..7',o[0,0,:8].tolist(),' direct',d[0,0,:8].tolist())
    print('  tiled row 100 head1',o[0,100,32:40].tolist(),' direct',d[0,100,32:40].tolist())
