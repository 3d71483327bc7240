import sys, os
sys.path[:0]=['/root/repo','/root/repo/generic-diffusion-feature_amd']
import torch
from components.native import NativeUNet, ARCH_CONFIGS
import bench as BB
ver=sys.argv[1] if len(sys.argv)>1 else '1-5'; B=int(sys.argv[2]) if len(sys.argv)>2 else 32
cfg=ARCH_CONFIGS[ver]; dev=torch.device('cuda:0'); lat=64 if ver=='1-5' else 128
import os
net=NativeUNet(cfg,device=dev,precise=(sys.argv[3] if len(sys.argv)>3 else False)).init_synthetic(0)
g=torch.Generator(device=dev).manual_seed(1)
x=torch.randn(B,4,lat,lat,generator=g,device=dev).half()
ctx=torch.randn(1,77,cfg['cross_attention_dim'],generator=g,device=dev).half().expand(B,-1,-1).contiguous()
t=torch.full((B,),100.0,device=dev); txt=tid=None
if cfg['addition_embed_text_time']:
    pooled=cfg['add_in_dim']-6*cfg['addition_time_embed_dim']
    txt=torch.randn(1,pooled,generator=g,device=dev).half().expand(B,-1).contiguous()
    tid=torch.tensor([[1024,1024,0,0,1024,1024]],dtype=torch.float32,device=dev).repeat(B,1)
ids=BB.PRACTICAL[ver]
for _ in range(3): net.forward_raw(x,t,ctx,txt,tid,hook_ids=ids,shared_ctx=True)
torch.cuda.synchronize()
acc={}
for rep in range(3):
    _,_,prof=net.forward_raw(x,t,ctx,txt,tid,hook_ids=ids,shared_ctx=True,profile=True)
    for i,(name,ms,fl,lab) in enumerate(prof):
        a=acc.setdefault(i,[name,0.0,fl,lab]); a[1]+=ms/3
tot=sum(a[1] for a in acc.values())
print('total ms',tot)
# group consecutive identical (name, flops, kernel)
grp={}
for i,(name,ms,fl,lab) in acc.items():
    k=(name,round(fl/1e9,1),lab) if len(sys.argv)<=4 else (name,0,''); g_=grp.setdefault(k,[0,0.0]); g_[0]+=1; g_[1]+=ms
for (name,gf,lab),(n,ms) in sorted(grp.items(), key=lambda kv:-kv[1][1])[:45]:
    print(f"{name:14s} n={n:3d} {ms:7.3f} ms  {gf:8.1f} GF/op  {gf*n/ms if ms>0 else 0:7.1f} TF  {lab}")
