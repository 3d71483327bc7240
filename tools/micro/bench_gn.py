#!/usr/bin/env python3
"""GroupNorm (stats + apply+SiLU) at the VAE / SDXL shapes through gdf_op_groupnorm: ms and effective HBM rate
(algorithmic bytes = read x twice + write y = 6 B per element)."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "tests"))
from ops_binding import P, lib, ok, stream
L = lib()
def t(fn, it=10):
    for _ in range(3): fn()
    torch.cuda.synchronize(); e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True); e0.record()
    for _ in range(it): fn()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) / it
for name, B, HW, C in (("vae128@1024", 4, 1024 * 1024, 128), ("vae256@512", 4, 512 * 512, 256), ("vae512@256", 4, 256 * 256, 512),
                       ("vae512@128", 4, 128 * 128, 512), ("xl320@128", 16, 128 * 128, 320), ("xl640@64", 16, 64 * 64, 640),
                       ("xl1280@32", 16, 32 * 32, 1280), ("xl2560@32", 16, 32 * 32, 2560)):
    x = torch.randn(B * HW, C, device="cuda").half(); y = torch.empty_like(x)
    gam = torch.ones(C, device="cuda"); bet = torch.zeros(C, device="cuda")
    scr = torch.empty(L.gdf_op_groupnorm_scratch_bytes(B, HW, C), dtype=torch.uint8, device="cuda")
    ms = t(lambda: ok(L.gdf_op_groupnorm(P(x), None, C, B, HW, C, 32, 1e-6, P(gam), P(bet), 1, P(y), P(scr), stream()), L))
    print(f"{name:14s} {ms:8.4f} ms  {6.0 * x.numel() / ms / 1e9:6.2f} TB/s (6 B/elem)")
