#!/usr/bin/env python3
"""copy2d_kernel (the hook-store kernel: strided fp16 / fp32 rows -> contiguous fp16 hook) through the C ABI: GB/s (read + write) by HIP events."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from ops_binding import P, lib, ok, stream
L = lib()
torch.cuda.set_device(0)
for (R, C, lds_, f32) in ((16384, 1280, 1280, 0), (16384, 1280, 2560, 0), (65536, 640, 640, 0), (16384, 1280, 1280, 1), (262144, 320, 320, 0), (4096, 1280, 1280, 0)):
    src = (torch.randn(R, lds_, device="cuda") if f32 else torch.randn(R, lds_, device="cuda").half())
    dst = torch.empty(R, C, dtype=torch.half, device="cuda")
    def run():
        ok(L.gdf_op_copy2d(None if f32 else P(src), P(src) if f32 else None, lds_, P(dst), C, R, C, stream()), L)
    for _ in range(5):
        run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = 200
    e0.record()
    for _ in range(n):
        run()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / n
    nbytes = R * C * (2 + (4 if f32 else 2))
    assert torch.equal(dst, src[:, :C].half())
    print(f"R={R:6d} C={C:4d} ld={lds_:4d} {'fp32' if f32 else 'fp16'} src: {us:7.2f} us  {nbytes / us / 1e3:7.1f} GB/s (read + write)  = {nbytes / us / 1e3 / 8000:.3f} of 8 TB/s")
