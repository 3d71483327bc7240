#!/usr/bin/env python3
"""A/B in one process: SDXL B=16 steps launched eagerly (default stream) vs replayed as hipGraphs (side stream)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "generic-diffusion-feature_amd")):
    sys.path.insert(0, p)
import torch
import bench
from components.native import NativeUNet
cfg = bench._cfg("xl"); B, lat = 16, 128
unet = NativeUNet(cfg, device="cuda:0").init_synthetic(0)
g = torch.Generator(device="cuda").manual_seed(1)
x = torch.randn(B, 4, lat, lat, generator=g, device="cuda").half()
ctx = torch.randn(1, 77, 2048, generator=g, device="cuda").half().expand(B, -1, -1).contiguous()
t = torch.full((B,), 100.0, device="cuda"); txt = torch.randn(1, 1280, generator=g, device="cuda").half().expand(B, -1).contiguous()
tid = torch.tensor([[1024, 1024, 0, 0, 1024, 1024]], dtype=torch.float32, device="cuda").repeat(B, 1)
ids = bench.PRACTICAL["xl"]
step = lambda: unet.forward_raw(x, t, ctx, txt, tid, hook_ids=ids, shared_ctx=True)
def run(n, stream=None):
    ctxm = torch.cuda.stream(stream) if stream is not None else torch.cuda.stream(torch.cuda.default_stream())
    with ctxm:
        for _ in range(3): o = step()
        torch.cuda.synchronize(); t0 = time.perf_counter(); c0 = time.process_time()
        for _ in range(n): o = step()
        c1 = time.process_time(); torch.cuda.synchronize(); dt = time.perf_counter() - t0
    return B * n / dt, (c1 - c0) / n * 1e3
side = torch.cuda.Stream()
for rnd in range(3):
    e, ce = run(10); gq, cg = run(10, side)
    print(f"round {rnd}: eager {e:7.2f} img/s (host {ce:5.2f} ms/step)   hipGraph {gq:7.2f} img/s (host {cg:5.2f} ms/step)")
