#!/usr/bin/env python3
"""PixArt-Sigma-XL-2 DiT at 1024x1024 (4096 image tokens, 300 caption tokens), batch 16, one MI355X: images/s + per-op profile.
    python tools/bench_pixart.py [--batch 16] [--steps 3]"""
import argparse, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "generic-diffusion-feature_amd")):
    sys.path.insert(0, p)
import torch
from components.native import NativePixArtTransformer, PIXART_CONFIGS
from oracle.pixart_ref import ARCH_PIXART_SIGMA, flops_per_image      # FLOP model only

ap = argparse.ArgumentParser(); ap.add_argument("--batch", type=int, default=16); ap.add_argument("--steps", type=int, default=3)
a = ap.parse_args()
cfg = PIXART_CONFIGS["pixart-sigma"]
net = NativePixArtTransformer(cfg, device="cuda:0").init_synthetic(0)
g = torch.Generator(device="cuda").manual_seed(1)
B, T = a.batch, 300
x = torch.randn(B, 4, 128, 128, device="cuda", generator=g).half()
enc = torch.randn(B, T, 4096, device="cuda", generator=g).half()
mask = (torch.arange(T, device="cuda")[None] < 120).expand(B, T).to(torch.int64)
t = torch.full((B,), 100.0, device="cuda")
ids = [f"vit-block{i}-out" for i in (6, 13, 20, 27)]
step = lambda **kw: net.forward_raw(x, enc, t, mask, hook_ids=ids, **kw)
step(); torch.cuda.synchronize()
_, _, prof = step(profile=True)
t0 = time.perf_counter()
for _ in range(a.steps):
    out = step()
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / a.steps
fl = flops_per_image(ARCH_PIXART_SIGMA, 4096, T)
assert torch.isfinite(out[0].float()).all()
print(json.dumps(dict(metric="images/sec feature-extract, PixArt-Sigma-XL-2 1024^2 single timestep", value=round(B / dt, 2), unit="images/s",
                      ms_per_step=round(dt * 1e3, 2), batch=B, tflop_per_image=round(fl / 1e12, 2), model_tflops_per_s=round(B * fl / dt / 1e12, 1),
                      hooks=len(out[1]), dtype="f16", data="synthetic")))
rows = {}
for name, ms, f_, k in prof:
    r = rows.setdefault(name, [0.0, 0.0, 0, k]); r[0] += ms; r[1] += f_; r[2] += 1
for name, r in sorted(rows.items(), key=lambda kv: -kv[1][0]):
    print(f"# {name:16s} n={r[2]:4d} {r[0]:9.3f} ms  {r[1] / 1e9 / max(r[0], 1e-9):8.1f} TFLOP/s  {r[3]}")
