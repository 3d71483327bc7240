#!/usr/bin/env python3
"""Kernel micro-benchmarks through the C ABI (include/gdf_ops.h) at the SDXL B=16 shapes.
    python tools/bench_ops.py [gemm] [conv] [attn] [norm]
"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
from ops_binding import P, lib, ok, stream  # noqa: E402

L = lib()
dev = "cuda"


def timeit(fn, iters=20, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def bench_gemm():
    shapes = [("ff_geglu", 16384, 10240, 1280, "geglu"), ("ff_geglu640", 65536, 5120, 640, "geglu"),
              ("ff_out", 16384, 1280, 5120, "res"), ("ff_out640", 65536, 640, 2560, "res"),
              ("qkv", 16384, 3840, 1280, ""), ("qkv640", 65536, 1920, 640, ""),
              ("attn_out", 16384, 1280, 1280, "res"), ("attn2_q", 16384, 1280, 1280, ""),
              ("attn_out640", 65536, 640, 640, "res"), ("kv", 1232, 2560, 2048, ""), ("shortcut", 16384, 1280, 2560, "o32"),
              ("square4k", 4096, 4096, 4096, ""), ("square8k", 8192, 8192, 8192, "")]
    print(f"{'gemm':14s} {'M':>6s} {'N':>6s} {'K':>5s} {'ms':>8s} {'TFLOP/s':>8s}")
    for name, M, N, K, mode in shapes:
        A = torch.randn(M, K, device=dev).half(); W = (torch.randn(N, K, device=dev) * K ** -0.5).half()
        bias = torch.randn(N, device=dev)
        No = N // 2 if mode == "geglu" else N
        o16 = torch.empty(M, No, device=dev, dtype=torch.half)
        o32 = torch.empty(M, No, device=dev) if mode in ("res", "o32") else None
        res = torch.randn(M, No, device=dev) if mode == "res" else None
        out = []
        for var in (128, 160, 256, 320, 825, 932):
            early = var >> 12; var &= 4095
            flags = (early << 20) | (var << 8) | ((1 | (8 if var in (160, 320) else 0)) if mode == "geglu" else 0)
            if (var in (160, 320) and N % var) or (var == 160 and mode == "geglu") or (var == 825 and (mode != "geglu" or N % 256)) or (var == 932 and (mode == "geglu" or N % 320)):
                out.append(" " * 17); continue
            fn = lambda: ok(L.gdf_op_gemm(P(A), K, P(W), P(bias), P(res), None, No, P(o16), No, P(o32), No, M, N, K,
                                          flags, stream()), L)
            ms = timeit(fn)
            out.append(f"{ms:8.4f} {2.0 * M * N * K / ms / 1e9:8.1f}")
        # library yardstick (hipBLASLt through torch.matmul, plain fp16 GEMM without our epilogue) — not a product path
        Wt = W.t().contiguous(); lib16 = torch.empty(M, N, device=dev, dtype=torch.half)
        ms_nt = timeit(lambda: torch.matmul(A, W.t(), out=lib16)); ms_nn = timeit(lambda: torch.matmul(A, Wt, out=lib16))
        ms_l = min(ms_nt, ms_nn)
        print(f"{name:14s} {M:6d} {N:6d} {K:5d} 128x128 {out[0]}  128x160 {out[1]}  256x128 {out[2]}  256x320 {out[3]}  8ph320 {out[4]}  8ph256 {out[5]}  8ph320b {out[6]}"
              f"  hipBLASLt {ms_l:8.4f} {2.0 * M * N * K / ms_l / 1e9:8.1f}")


def bench_conv():
    shapes = [("320@128", 16, 128, 128, 320, 320, 1, 0), ("640->320@128", 16, 128, 128, 640, 320, 1, 0),
              ("960->320@128", 16, 128, 128, 960, 320, 1, 0), ("640@64", 16, 64, 64, 640, 640, 1, 0),
              ("1920->640@64", 16, 64, 64, 1920, 640, 1, 0), ("1280@32", 16, 32, 32, 1280, 1280, 1, 0),
              ("2560->1280@32", 16, 32, 32, 2560, 1280, 1, 0), ("up640@64->128", 16, 64, 64, 640, 640, 1, 1),
              ("down320@128", 16, 128, 128, 320, 320, 2, 0)]
    print(f"{'conv':16s} {'ms':>8s} {'TFLOP/s':>8s}")
    for name, B, H, W, Ci, Co, st, ups in shapes:
        x = torch.randn(B, H, W, Ci, device=dev).half(); w = (torch.randn(Co, 9 * Ci, device=dev) * (9 * Ci) ** -0.5).half()
        bias = torch.randn(Co, device=dev)
        OH = (2 * H if ups else H) // st; OW = (2 * W if ups else W) // st
        o16 = torch.empty(B, OH, OW, Co, device=dev, dtype=torch.half)
        out = []
        for var in (128, 160, 256, 320, 932):
            if var == 932 and Co % 320:
                out.append(" " * 17); continue
            fn = lambda: ok(L.gdf_op_conv3x3(P(x), Ci, B, H, W, Ci, P(w), Co, P(bias), None, st, ups, None, None, P(o16), None,
                                             ((var >> 12) << 20) | ((var & 4095) << 8), stream()), L)
            ms = timeit(fn, iters=10)
            out.append(f"{ms:8.4f} {2.0 * B * OH * OW * Co * 9 * Ci / ms / 1e9:8.1f}")
        print(f"{name:16s} 128x128 {out[0]}  128x160 {out[1]}  256x128 {out[2]}  256x320 {out[3]}  8ph320b {out[4]}")


def bench_attn():
    shapes = [("self1024", 16, 20, 1024, 1024, 64), ("self4096", 16, 10, 4096, 4096, 64), ("cross1024", 16, 20, 1024, 77, 64),
              ("cross4096", 16, 10, 4096, 77, 64), ("sd15_4096_d40", 8, 8, 4096, 4096, 40), ("sd15_1024_d80", 8, 8, 1024, 1024, 80),
              ("sd15_256_d160", 8, 8, 256, 256, 160)]
    print(f"{'attn':16s} {'ms':>8s} {'TFLOP/s':>8s}")
    for name, B, h, Sq, Sk, D in shapes:
        C = h * D
        self_attn = Sq == Sk
        if self_attn:
            qkv = torch.randn(B * Sq, 3 * C, device=dev).half()
            q, k, v, ld = qkv, qkv[:, C:], qkv[:, 2 * C:], 3 * C
            args = (P(q), ld, torch_ptr(qkv, C), ld, torch_ptr(qkv, 2 * C), ld)
        else:
            q = torch.randn(B * Sq, C, device=dev).half(); kv = torch.randn(B * Sk, 2 * C, device=dev).half()
            args = (P(q), C, P(kv), 2 * C, torch_ptr(kv, C), 2 * C)
        o = torch.empty(B * Sq, C, device=dev, dtype=torch.half)
        fn = lambda: ok(L.gdf_op_attention(*args, P(o), C, B, h, Sq, Sk, D, None, stream()), L)
        ms = timeit(fn, iters=10)
        print(f"{name:16s} {ms:8.4f} {4.0 * B * h * Sq * Sk * D / ms / 1e9:8.1f}")


def torch_ptr(t, col):
    import ctypes
    return ctypes.c_void_p(t.data_ptr() + col * t.element_size())


def bench_norm():
    print(f"{'norm':22s} {'ms':>8s} {'GB/s':>8s}")
    for name, R, C in [("ln1280x16384", 16384, 1280), ("ln640x65536", 65536, 640)]:
        x = torch.randn(R, C, device=dev); g = torch.ones(C, device=dev); b = torch.zeros(C, device=dev)
        y = torch.empty(R, C, device=dev, dtype=torch.half)
        fn = lambda: ok(L.gdf_op_layernorm(None, P(x), C, R, C, 1e-5, P(g), P(b), P(y), stream()), L)
        ms = timeit(fn)
        print(f"{name:22s} {ms:8.4f} {R * C * 6 / ms / 1e6:8.1f}")
    for name, B, HW, C in [("gn320@128", 16, 16384, 320), ("gn960@128", 16, 16384, 960), ("gn1280@32", 16, 1024, 1280),
                           ("gn2560@32", 16, 1024, 2560)]:
        x = torch.randn(B, HW, C, device=dev).half(); g = torch.ones(C, device=dev); b = torch.zeros(C, device=dev)
        y = torch.empty(B, HW, C, device=dev, dtype=torch.half)
        sc = torch.empty(L.gdf_op_groupnorm_scratch_bytes(B, HW, C) + 1024, device=dev, dtype=torch.uint8)
        fn = lambda: ok(L.gdf_op_groupnorm(P(x), None, C, B, HW, C, 32, 1e-5, P(g), P(b), 1, P(y), P(sc), stream()), L)
        ms = timeit(fn)
        print(f"{name:22s} {ms:8.4f} {B * HW * C * 6 / ms / 1e6:8.1f}   (read x twice + write y)")


if __name__ == "__main__":
    which = sys.argv[1:] or ["gemm", "conv", "attn", "norm"]
    for w in which:
        {"gemm": bench_gemm, "conv": bench_conv, "attn": bench_attn, "norm": bench_norm}[w]()
