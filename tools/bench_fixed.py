#!/usr/bin/env python3
"""Per-round fixed cost of the two-group GEMM tiles: tiny-K launches (1, 2, 4 K-tiles) with and without the output stores."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
from ops_binding import P, lib, ok, stream
from bench_ops import timeit
L = lib(); dev = "cuda"
for (name, M, N, geglu, var, rounds) in [("geglu 825", 16384, 10240, 1, 825, 10), ("plain 932", 16384, 3840, 0, 932, 3), ("plain 932", 16384, 1280, 0, 932, 1),
                                         ("geglu 825", 1024, 10240, 1, 825, 160 / 256.0)]:
    for K in (64, 128, 256, 512):
        A = torch.randn(M, K, device=dev).half(); W = (torch.randn(N, K, device=dev) * K ** -0.5).half()
        bias = torch.randn(N, device=dev); No = N // 2 if geglu else N
        o16 = torch.empty(M, No, device=dev, dtype=torch.half)
        flags = (var << 8) | geglu
        f1 = lambda: ok(L.gdf_op_gemm(P(A), K, P(W), P(bias), None, None, No, P(o16), No, None, No, M, N, K, flags, stream()), L)
        f0 = lambda: ok(L.gdf_op_gemm(P(A), K, P(W), P(bias), None, None, No, None, No, None, No, M, N, K, flags, stream()), L)
        t1, t0 = timeit(f1), timeit(f0)
        print(f"{name} {M}x{N}x{K}: {t1*1e3:7.1f} us ({t1*1e3/max(rounds,1):6.2f} us/round) | no store {t0*1e3:7.1f} us ({t0*1e3/max(rounds,1):6.2f} us/round)")
