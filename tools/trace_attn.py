#!/usr/bin/env python3
"""Shader-clock time per loop phase of the attention kernel (tools/ablate_attn.sh builds the stamped library): wave 0 of every
workgroup sums [QK^T | softmax | PV | stage + barrier + next loads] over its K/V tiles.   python tools/trace_attn.py"""
import ctypes as C, os, sys, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
vp, ci = C.c_void_p, C.c_int
L = C.CDLL(os.path.join(ROOT, "tools/micro/build", f"libgdf_attn_trace{sys.argv[1] if len(sys.argv) > 1 else 0}.so"))
L.gdf_op_attention.restype = ci
L.gdf_op_attention.argtypes = [vp, ci, vp, ci, vp, ci, vp, ci, ci, ci, ci, ci, ci, vp, vp]
_w = torch.randn(8192, 8192, device="cuda").half()
for _ in range(100): _w @ _w
for name, B, h, S, D, qblk, Sk in [("sdxl_4096", 16, 10, 4096, 64, 256, 4096), ("sdxl_1024", 16, 20, 1024, 64, 256, 1024), ("sdxl_cross_1024x77", 16, 20, 1024, 64, 256, 77), ("sdxl_cross_4096x77", 16, 10, 4096, 64, 256, 77)]:
    Cw = h * D
    qkv = torch.randn(B * S, 3 * Cw, device="cuda").half(); o = torch.empty(B * S, Cw, device="cuda", dtype=torch.half)
    kv = torch.randn(B * Sk, 2 * Cw, device="cuda").half()
    s = vp(torch.cuda.current_stream().cuda_stream)
    if Sk == S: fn = lambda: L.gdf_op_attention(vp(qkv.data_ptr()), 3 * Cw, vp(qkv.data_ptr() + Cw * 2), 3 * Cw, vp(qkv.data_ptr() + 4 * Cw), 3 * Cw, vp(o.data_ptr()), Cw, B, h, S, S, D, None, s)
    else: fn = lambda: L.gdf_op_attention(vp(qkv.data_ptr()), 3 * Cw, vp(kv.data_ptr()), 2 * Cw, vp(kv.data_ptr() + Cw * 2), 2 * Cw, vp(o.data_ptr()), Cw, B, h, S, Sk, D, None, s)
    for _ in range(3): assert fn() == 0
    torch.cuda.synchronize()
    nwg = min(8192, B * h * (S // qblk))
    buf = np.zeros(nwg * 8, dtype=np.uint64)
    assert L.gdf_debug_attn_trace(buf.ctypes.data_as(C.POINTER(C.c_ulonglong)), nwg * 8) == 0
    t = buf.reshape(nwg, 8).astype(np.float64); nt = (Sk + 63) // 64
    ph = t[:, :4].mean(0) / nt
    print(f"{name}: per tile (cycles, mean over {nwg} workgroups): QK^T {ph[0]:7.0f}  softmax {ph[1]:7.0f}  PV {ph[2]:7.0f}  stage+barrier+loads {ph[3]:7.0f}  "
          f"sum {ph.sum():7.0f}   | whole loop {t[:, 4].mean():9.0f} cycles = {t[:, 4].mean() / nt:7.0f} per tile | entry -> loop {t[:, 5].mean():7.0f}, "
          f"whole workgroup {t[:, 6].mean():8.0f} cycles")
