#!/usr/bin/env python3
"""A handful of GEMM / attention launches for rocprofv3 --pmc runs (kept tiny so counter output stays small)."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
from ops_binding import P, lib, ok, stream
import ctypes
L = lib(); dev = "cuda"
which = sys.argv[1:] or ["gemm"]
if "gemm" in which:
    for (M, N, K, flags) in [(8192, 8192, 8192, 0), (16384, 10240, 1280, 1), (16384, 1280, 1280, 0), (16384, 1280, 1280, 4)]:
        A = torch.randn(M, K, device=dev).half(); W = (torch.randn(N, K, device=dev) * K ** -0.5).half()
        bias = torch.randn(N, device=dev); No = N // 2 if flags & 1 else N
        o16 = torch.empty(M, No, device=dev, dtype=torch.half)
        for _ in range(3):
            ok(L.gdf_op_gemm(P(A), K, P(W), P(bias), None, None, No, P(o16), No, None, No, M, N, K, flags, stream()), L)
        torch.cuda.synchronize()
if "attn" in which:
    for (B, h, S, D) in [(16, 20, 1024, 64), (16, 10, 4096, 64)]:
        C = h * D
        qkv = torch.randn(B * S, 3 * C, device=dev).half(); o = torch.empty(B * S, C, device=dev, dtype=torch.half)
        pk = ctypes.c_void_p(qkv.data_ptr() + C * 2); pv = ctypes.c_void_p(qkv.data_ptr() + 2 * C * 2)
        for _ in range(3):
            ok(L.gdf_op_attention(P(qkv), 3 * C, pk, 3 * C, pv, 3 * C, P(o), C, B, h, S, S, D, None, stream()), L)
        torch.cuda.synchronize()
