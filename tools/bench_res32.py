#!/usr/bin/env python3
"""fp32-residual GEMMs of the SDXL step (attention out-projections, ff_out) on the 256x320 two-group tile vs the 128x160 ring."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
from ops_binding import P, lib, ok, stream
from bench_ops import timeit
L = lib(); dev = "cuda"
for (name, M, N, K) in [("attn_out L2", 16384, 1280, 1280), ("attn_out L1", 65536, 640, 640), ("ff_out L2", 16384, 1280, 5120), ("ff_out L1", 65536, 640, 2560),
                        ("attn_out SD1.5 L0 B32", 131072, 320, 320), ("attn_out SD1.5 L1 B32", 32768, 640, 640), ("attn_out SD1.5 L2 B32", 8192, 1280, 1280)]:
    A = torch.randn(M, K, device=dev).half(); W = (torch.randn(N, K, device=dev) * K ** -0.5).half(); bias = torch.randn(N, device=dev)
    o32 = torch.empty(M, N, device=dev); res = torch.randn(M, N, device=dev)
    row = []
    for var in (932, 160, 320, 0):
        fn = lambda: ok(L.gdf_op_gemm(P(A), K, P(W), P(bias), P(res), None, N, None, N, P(o32), N, M, N, K, var << 8, stream()), L)
        t = timeit(fn)
        row.append(f"{var or 'auto':>4}: {t*1e3:7.1f} us {2.0*M*N*K/t/1e9:7.1f} TF")
    print(f"{name:24s} {M}x{N}x{K}: " + " | ".join(row), flush=True)
