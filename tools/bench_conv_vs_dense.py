#!/usr/bin/env python3
"""What bounds the narrow-N VAE convs?  The same tile kernels on (a) the 3x3 conv as implicit GEMM (every A element staged 9 times from L2) and (b) a
DENSE GEMM of the same M x N x K (every A element staged once): if (b) is not faster, the im2col re-staging is not the bound and an LDS halo-tile conv
would not pay.    python tools/bench_conv_vs_dense.py"""
import sys, os, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from ops_binding import P, lib, ok, stream
L = lib()
def t(fn, it=5):
    for _ in range(2): fn()
    torch.cuda.synchronize(); e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True); e0.record()
    for _ in range(it): fn()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) / it
for name, B, H, Ci, Co in (("128@512 B=3", 3, 512, 128, 128), ("128@1024", 4, 1024, 128, 128), ("256@512", 4, 512, 256, 256), ("512@256", 4, 256, 512, 512), ("320@128 (SDXL)", 16, 128, 320, 320), ("640@64 (SDXL)", 16, 64, 640, 640)):
    M, K = B * H * H, 9 * Ci
    x = torch.randn(B, H, H, Ci, device="cuda").half(); w = (torch.randn(Co, K, device="cuda") * K ** -0.5).half()
    bias = torch.randn(Co, device="cuda"); o16 = torch.empty(M, Co, device="cuda", dtype=torch.half)
    a = torch.randn(M, K, device="cuda").half() if M * K * 2 < (1 << 31) else None
    row = []
    for var in (128, 256, 826, 932):
        if (var == 826 and Co % 256) or (var == 932 and Co % 320): row.append("          -          "); continue
        mc = t(lambda: ok(L.gdf_op_conv3x3(P(x), Ci, B, H, H, Ci, P(w), Co, P(bias), None, 1, 0, None, None, P(o16), None, var << 8, stream()), L))
        md = t(lambda: ok(L.gdf_op_gemm(P(a), K, P(w), P(bias), None, None, 0, P(o16), Co, None, 0, M, Co, K, (932 if var == 932 else var) << 8, stream()), L)) if a is not None and var != 826 else float("nan")
        fl = 2.0 * M * Co * K / 1e9
        row.append(f"conv {fl / mc:5.0f} dense {fl / md:5.0f}")
    print(f"{name:16s} 128x128 {row[0]} | 256x128 {row[1]} | 8ph256x256 {row[2]} | 8ph256x320 {row[3]}   (TFLOP/s)")
