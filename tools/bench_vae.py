#!/usr/bin/env python3
"""VAE encode + sample + noise-add (include/gdf_vae.h) at the headline shape: 16 images of 1024x1024, per-op profile.
    python tools/bench_vae.py [--batch 16] [--img 1024] [--steps 3]"""
import argparse, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "generic-diffusion-feature_amd")):
    sys.path.insert(0, p)
import torch
from components.native import NativeVAEEncoder, VAE_CONFIGS
from oracle.vae_ref import ARCH_SD_VAE, flops_per_image     # FLOP model only

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=16); ap.add_argument("--img", type=int, default=1024)
ap.add_argument("--steps", type=int, default=3)
a = ap.parse_args()
enc = NativeVAEEncoder(VAE_CONFIGS["sd"], device="cuda:0").init_synthetic(0)
g = torch.Generator(device="cuda").manual_seed(0)
x = (torch.rand(a.batch, 3, a.img, a.img, device="cuda", generator=g) * 2 - 1).half()
eps = torch.randn(a.batch, 4, a.img // 8, a.img // 8, device="cuda", generator=g).half(); noise = torch.randn_like(eps)
kw = dict(eps=eps, noise=noise, scaling_factor=0.13025, noise_a=1.0, noise_b=0.6, input_scale=0.86)
enc.encode(x, **kw); torch.cuda.synchronize()
_, prof = enc.encode(x, profile=True, **kw)
t0 = time.perf_counter()
for _ in range(a.steps):
    out = enc.encode(x, **kw)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / a.steps
fl = flops_per_image(ARCH_SD_VAE, a.img)
plan = enc._plan(a.batch, a.img, a.img)
print(f"VAE encode+sample+noise: batch {a.batch} x {a.img}^2: {dt * 1e3:.1f} ms/batch = {a.batch / dt:.1f} img/s, "
      f"{fl / 1e12:.2f} TFLOP/img -> {a.batch * fl / dt / 1e12:.0f} TFLOP/s; workspace {plan.ws_bytes / 1e9:.2f} GB; finite={bool(torch.isfinite(out.float()).all())}")
rows = {}
for name, ms, f_, k in prof:
    r = rows.setdefault(name, [0.0, 0.0, 0]); r[0] += ms; r[1] += f_; r[2] += 1
print("# per-op profile of ONE sub-batch pass:")
for name, r in sorted(rows.items(), key=lambda kv: -kv[1][0]):
    print(f"# {name:18s} n={r[2]:4d} {r[0]:9.3f} ms  {r[1] / 1e9 / max(r[0], 1e-9):8.1f} TFLOP/s")

# ---- `vae-out`: scheduler step + decoder (include/gdf_vae.h decoder half) at the same shape ----
from components.native import NativeVAEDecoder
from oracle.vae_ref import dec_flops_per_image
dec = NativeVAEDecoder(VAE_CONFIGS["sd"], device="cuda:0").init_synthetic(1)
lat = torch.randn(a.batch, 4, a.img // 8, a.img // 8, device="cuda", generator=g).half(); npred = torch.randn_like(lat)
dkw = dict(c_sample=1.0, c_eps=-0.4, scaling_factor=0.13025)
dec.decode(lat, npred, **dkw); torch.cuda.synchronize()
_, dprof = dec.decode(lat, npred, profile=True, **dkw)
t0 = time.perf_counter()
for _ in range(a.steps):
    img = dec.decode(lat, npred, **dkw)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / a.steps
fl = dec_flops_per_image(ARCH_SD_VAE, a.img)
dplan = dec._plan(a.batch, a.img // 8, a.img // 8)
print(f"VAE step+decode (vae-out): batch {a.batch} x {a.img}^2: {dt * 1e3:.1f} ms/batch = {a.batch / dt:.1f} img/s, "
      f"{fl / 1e12:.2f} TFLOP/img -> {a.batch * fl / dt / 1e12:.0f} TFLOP/s; workspace {dplan.ws_bytes / 1e9:.2f} GB; finite={bool(torch.isfinite(img.float()).all())}")
rows = {}
for name, ms, f_, k in dprof:
    r = rows.setdefault(name, [0.0, 0.0, 0]); r[0] += ms; r[1] += f_; r[2] += 1
print("# per-op profile of ONE sub-batch pass (decoder):")
for name, r in sorted(rows.items(), key=lambda kv: -kv[1][0]):
    print(f"# {name:18s} n={r[2]:4d} {r[0]:9.3f} ms  {r[1] / 1e9 / max(r[0], 1e-9):8.1f} TFLOP/s")
