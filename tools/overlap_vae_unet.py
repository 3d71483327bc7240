#!/usr/bin/env python3
"""Does running the VAE encode of batch i+1 on a second stream beside the UNet forward of batch i buy the product call anything?
SDXL 1024^2, batch 16, practical hooks: serial (VAE, UNet, VAE, UNet ... on one stream) against two streams (the VAE of the next batch is queued on
stream B while the UNet of the current one runs on stream A; the UNet waits for ITS latents through an event).  img/s of the pair.
    python tools/overlap_vae_unet.py [--pairs 6]"""
import argparse, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "generic-diffusion-feature_amd")):
    sys.path.insert(0, p)
import torch
from components.native import NativeVAEEncoder, NativeUNet, VAE_CONFIGS
import bench

ap = argparse.ArgumentParser(); ap.add_argument("--pairs", type=int, default=6); ap.add_argument("--batch", type=int, default=16)
a = ap.parse_args()
dev = torch.device("cuda:0"); B = a.batch
cfg = bench._cfg("xl")
unet = NativeUNet(cfg, device=dev); unet.init_synthetic(seed=0)
ids = list(json.load(open(os.path.join(ROOT, "generic-diffusion-feature_amd", "configs", "config_xl_practical.json"))).keys())
enc = NativeVAEEncoder(VAE_CONFIGS["sd"], device="cuda:0").init_synthetic(0)
g = torch.Generator(device="cuda").manual_seed(0)
x = (torch.rand(B, 3, 1024, 1024, device="cuda", generator=g) * 2 - 1).half()
eps = torch.randn(B, 4, 128, 128, device="cuda", generator=g).half(); noise = torch.randn_like(eps)
kw = dict(eps=eps, noise=noise, scaling_factor=0.13025, noise_a=1.0, noise_b=0.6, input_scale=0.86)
ctx = torch.randn(1, 77, 2048, device="cuda", generator=g).half().expand(B, -1, -1).contiguous()
txt = torch.randn(B, 1280, device="cuda", generator=g).half(); tid = torch.tensor([[1024, 1024, 0, 0, 1024, 1024]] * B, device="cuda", dtype=torch.float32)
t = torch.full((B,), 100.0)

def fwd(lat):
    return unet.forward_raw(lat, t, ctx, txt, tid, hook_ids=ids, shared_ctx=True)

for _ in range(3):
    lat = enc.encode(x, **kw); out = fwd(lat)
torch.cuda.synchronize()

def serial(n):
    t0 = time.perf_counter()
    for _ in range(n):
        lat = enc.encode(x, **kw); out = fwd(lat)
    torch.cuda.synchronize()
    return B * n / (time.perf_counter() - t0)

sA, sB = torch.cuda.Stream(), torch.cuda.Stream()
def overlapped(n):
    t0 = time.perf_counter()
    with torch.cuda.stream(sB):
        lat = enc.encode(x, **kw); ev = torch.cuda.Event(); ev.record()
    outs = []
    for i in range(n):
        with torch.cuda.stream(sA):
            sA.wait_event(ev)
            cur = lat
            out = fwd(cur); outs.append(out); done = torch.cuda.Event(); done.record()
        if i + 1 < n:
            with torch.cuda.stream(sB):          # the next batch's VAE: queued while the UNet above runs
                lat = enc.encode(x, **kw); ev = torch.cuda.Event(); ev.record()
        if len(outs) > 2:
            outs.pop(0)
    torch.cuda.synchronize()
    return B * n / (time.perf_counter() - t0)

for r in range(3):
    print(f"run {r}: serial {serial(a.pairs):6.2f} img/s   two streams {overlapped(a.pairs):6.2f} img/s", flush=True)
