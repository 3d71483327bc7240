#!/usr/bin/env python3
"""SD1.5's deepest level (8x8 latent, B = 32: M = 2048 output pixels, N = 1280, K = 9 x 1280 .. 9 x 2560) and the 16x16 level (M = 8192): which
tile / split-K factor runs these few-tile, long-K convs fastest?   python tools/bench_conv_small_m.py   (GDF_SPLITK_TILE=160 for the 128x160 split tile)"""
import sys, os, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from ops_binding import P, lib, ok, stream
import ctypes as C
L = lib()
L.gdf_op_conv3x3_splitk.restype = C.c_int
L.gdf_op_conv3x3_splitk.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_int,
                                    C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
L.gdf_op_splitk_factor.restype = C.c_int
L.gdf_op_splitk_factor.argtypes = [C.c_int] * 4
def t(fn, it=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True); e0.record()
    for _ in range(it): fn()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) / it
for name, B, H, Ci, Co in (("1280->1280 @8x8 B=32", 32, 8, 1280, 1280), ("2560->1280 @8x8 B=32", 32, 8, 2560, 1280), ("1280->1280 @16x16 B=32", 32, 16, 1280, 1280),
                           ("2560->1280 @16x16 B=32", 32, 16, 2560, 1280), ("1920->1280 @16x16", 32, 16, 1920, 1280), ("640->640 @32x32 B=32", 32, 32, 640, 640)):
    M, K = B * H * H, 9 * Ci
    x = torch.randn(B, H, H, Ci, device="cuda").half(); w = (torch.randn(Co, K, device="cuda") * K ** -0.5).half()
    bias = torch.randn(Co, device="cuda"); o16 = torch.empty(M, Co, device="cuda", dtype=torch.half)
    fl = 2.0 * M * Co * K / 1e9
    row = [f"{name:24s} M={M:5d} chosen split-K {L.gdf_op_splitk_factor(M, Co, K, 1)}:"]
    for var in (128, 160, 932):
        if (var == 932 and Co % 320) or (var == 160 and Co % 160): continue
        ms = t(lambda: ok(L.gdf_op_conv3x3(P(x), Ci, B, H, H, Ci, P(w), Co, P(bias), None, 1, 0, None, None, P(o16), None, var << 8, stream()), L))
        row.append(f"v{var} {fl / ms:5.0f}")
    ws = torch.empty(8 * M * Co, device="cuda")
    for sk in (2, 3, 4, 6, 8):
        ms = t(lambda: ok(L.gdf_op_conv3x3_splitk(P(x), Ci, B, H, H, Ci, P(w), Co, P(bias), None, 1, 0, None, None, P(o16), None, sk, P(ws), stream()), L))
        row.append(f"sk{sk} {fl / ms:5.0f}")
    print("  ".join(row), "  (TFLOP/s)")
