import os, sys
ROOT="/root/repo"
for p in (ROOT, os.path.join(ROOT, "generic-diffusion-feature_amd")): sys.path.insert(0, p)
import torch
from components.native import NativeVAEEncoder, VAE_CONFIGS
B=int(sys.argv[1]) if len(sys.argv)>1 else 16; img=int(sys.argv[2]) if len(sys.argv)>2 else 1024
enc = NativeVAEEncoder(VAE_CONFIGS["sd"], device="cuda:0").init_synthetic(0)
g = torch.Generator(device="cuda").manual_seed(0)
x = (torch.rand(B, 3, img, img, device="cuda", generator=g) * 2 - 1).half()
eps = torch.randn(B, 4, img // 8, img // 8, device="cuda", generator=g).half(); noise = torch.randn_like(eps)
kw = dict(eps=eps, noise=noise, scaling_factor=0.13025, noise_a=1.0, noise_b=0.6, input_scale=0.86)
enc.encode(x, **kw); torch.cuda.synchronize()
acc=None
for r in range(3):
    _, prof = enc.encode(x, profile=True, **kw)
    if acc is None: acc=[[n,0.0,f,k] for n,ms,f,k in prof]
    for a,(n,ms,f,k) in zip(acc,prof): a[1]+=ms/3
tot=sum(a[1] for a in acc)
print(f"ops {len(acc)} total {tot:.2f} ms (one sub-batch)")
for n,ms,f,k in acc:
    print(f"{n:18s} {ms:7.3f} ms {f/1e9:9.1f} GF {f/1e9/ms if ms>0 and f>0 else 0:7.1f} TF  {k}")
