#!/usr/bin/env python3
"""SD1.5 B = 32 GEMM shapes per tile variant (TFLOP/s): the GEGLU projections (K = 320 / 640 / 1280) and the plain dense GEMMs of level 0.
    python tools/bench_gemm_shapes_sd15.py"""
import sys, os, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from ops_binding import P, lib, ok, stream
L = lib()
def t(fn, it=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True); e0.record()
    for _ in range(it): fn()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) / it
print("GEGLU projection (M, N = 8C, K = C), variants 0 = the picker's choice, 825 = 8-phase 256x256, 320 = 256x320 ring, 256 = 256x128 ring, 128 = 128x128 ring")
for M, C in ((131072, 320), (32768, 640), (8192, 1280), (2048, 1280)):
    x = torch.randn(M, C, device="cuda").half(); W = (torch.randn(8 * C, C, device="cuda") * C ** -0.5).half(); b = torch.randn(8 * C, device="cuda")
    Wd = torch.empty_like(W); bd = torch.empty_like(b)
    ok(L.gdf_op_relayout_geglu(P(W), P(b), P(Wd), P(bd), 8 * C, C, 16, stream()), L)
    out = torch.empty(M, 4 * C, dtype=torch.half, device="cuda")
    fl = 2.0 * M * 8 * C * C / 1e9
    row = []
    for var in (0, 825, 320, 256, 128, 0, 825):
        if var == 320 and (8 * C // 2) % 320: row.append("   -  "); continue
        rc = L.gdf_op_gemm(P(x), C, P(Wd), P(bd), None, None, 0, P(out), 4 * C, None, 0, M, 8 * C, C, 1 | (var << 8), stream())
        torch.cuda.synchronize()
        if rc != 0: row.append("  n/a "); continue
        ms = t(lambda: L.gdf_op_gemm(P(x), C, P(Wd), P(bd), None, None, 0, P(out), 4 * C, None, 0, M, 8 * C, C, 1 | (var << 8), stream()))
        row.append(f"v{var}: {fl / ms:5.0f} ({ms * 1e3:4.0f} us)")
    print(f"M={M:6d} C={C:4d}: " + "  ".join(row))
print("dense GEMMs of SD1.5 level 0 / 1 (fp16 out): qkv (M, 3C, C), ff_out-like (M, C, 4C) with fp32 residual in / out")
for M, C in ((131072, 320), (32768, 640)):
    x = torch.randn(M, C, device="cuda").half(); W = (torch.randn(3 * C, C, device="cuda") * C ** -0.5).half(); b = torch.randn(3 * C, device="cuda")
    out = torch.empty(M, 3 * C, dtype=torch.half, device="cuda")
    fl = 2.0 * M * 3 * C * C / 1e9
    row = []
    for var in (0, 932, 160, 256, 128):
        rc = L.gdf_op_gemm(P(x), C, P(W), P(b), None, None, 0, P(out), 3 * C, None, 0, M, 3 * C, C, var << 8, stream())
        torch.cuda.synchronize()
        if rc != 0: row.append("  n/a "); continue
        ms = t(lambda: L.gdf_op_gemm(P(x), C, P(W), P(b), None, None, 0, P(out), 3 * C, None, 0, M, 3 * C, C, var << 8, stream()))
        row.append(f"v{var}: {fl / ms:5.0f} ({ms * 1e3:4.0f} us)")
    print(f"qkv    M={M:6d} C={C:4d}: " + "  ".join(row))
    x4 = torch.randn(M, 4 * C, device="cuda").half(); W2 = (torch.randn(C, 4 * C, device="cuda") * (4 * C) ** -0.5).half(); b2 = torch.randn(C, device="cuda")
    res = torch.randn(M, C, device="cuda"); o32 = torch.empty(M, C, device="cuda")
    fl = 2.0 * M * C * 4 * C / 1e9
    row = []
    for var in (0, 932, 160, 128):
        rc = L.gdf_op_gemm(P(x4), 4 * C, P(W2), P(b2), P(res), None, C, None, 0, P(o32), C, M, C, 4 * C, var << 8, stream())
        torch.cuda.synchronize()
        if rc != 0: row.append("  n/a "); continue
        ms = t(lambda: L.gdf_op_gemm(P(x4), 4 * C, P(W2), P(b2), P(res), None, C, None, 0, P(o32), C, M, C, 4 * C, var << 8, stream()))
        row.append(f"v{var}: {fl / ms:5.0f} ({ms * 1e3:4.0f} us)")
    print(f"ff_out M={M:6d} C={C:4d}: " + "  ".join(row))
