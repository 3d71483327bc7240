#!/usr/bin/env python3
"""Fixed cost (prologue+epilogue) vs main-loop rate of the GEMM variants: time vs K at fixed M,N."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
from ops_binding import P, lib, ok, stream
from bench_ops import timeit
L = lib(); dev = "cuda"
for (name, M, N, mode, var) in [("geglu 825 (256x256 two-group)", 16384, 10240, "geglu", 825), ("geglu 320 ring", 16384, 10240, "geglu", 320),
                                ("plain 932 N3840 (3 rounds)", 16384, 3840, "", 932), ("plain 932 N1280 (1 round)", 16384, 1280, "", 932),
                                ("res32 932 N1280", 16384, 1280, "res", 932), ("res32 160 N1280", 16384, 1280, "res", 160),
                                ("geglu 825 M65536 N5120", 65536, 5120, "geglu", 825), ("plain 932 M65536 N640", 65536, 640, "", 932)]:
    rows = []
    for K in (640, 1280, 2560, 5120):
        A = torch.randn(M, K, device=dev).half(); W = (torch.randn(N, K, device=dev) * K ** -0.5).half()
        bias = torch.randn(N, device=dev); No = N // 2 if mode == "geglu" else N
        o16 = torch.empty(M, No, device=dev, dtype=torch.half)
        o32 = torch.empty(M, No, device=dev) if mode == "res" else None
        res = torch.randn(M, No, device=dev) if mode == "res" else None
        flags = (var << 8) | (1 if mode == "geglu" else 0)
        fn = lambda: ok(L.gdf_op_gemm(P(A), K, P(W), P(bias), P(res), None, No, P(o16), No, P(o32), No, M, N, K, flags, stream()), L)
        rows.append((K, timeit(fn)))
    (k0, t0), (k1, t1) = rows[1], rows[3]
    slope = (t1 - t0) / (k1 - k0)                      # ms per unit K
    fixed = t0 - slope * k0
    rate = 2.0 * M * N / slope / 1e9
    print(f"{name:30s} " + " ".join(f"K{k}:{t:.4f}ms" for k, t in rows) + f"  main-loop {rate:7.1f} TFLOP/s, fixed {fixed*1e3:7.1f} us/launch")
