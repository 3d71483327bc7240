#!/usr/bin/env python3
"""GEMM micro-benchmark through gdf_op_gemm_dit: tile variants side by side at the Flux / PixArt / square shapes.
    python tools/bench_gemm_dit.py [variant ...]      (default: 1256 8256)"""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
from ops_binding import P, lib, ok, stream
L = lib()
variants = [int(a) for a in sys.argv[1:]] or [1256, 8256]
def t(fn, it=10):
    for _ in range(3): fn()
    torch.cuda.synchronize(); e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True); e0.record()
    for _ in range(it): fn()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) / it
shapes = [("flux_qkv", 36864, 9216, 3072), ("flux_proj_mlp", 36864, 12288, 3072), ("flux_proj_out", 36864, 3072, 15360),
          ("flux_ff_out", 32768, 3072, 12288), ("sdxl_ff1", 16384, 10240, 1280), ("sdxl_qkv", 16384, 3840, 1280),
          ("sdxl_ff_out", 16384, 1280, 5120), ("pixart_qkv", 65536, 3456, 1152), ("pixart_out", 65536, 1152, 1152),
          ("pixart_ff_out", 65536, 1152, 4608), ("pixart_ff_in", 65536, 4608, 1152), ("square4k", 4096, 4096, 4096), ("square8k", 8192, 8192, 8192)]
for name, M, N, K in shapes:
    A = torch.randn(M, K, device="cuda").half(); W = (torch.randn(N, K, device="cuda") * K ** -0.5).half()
    bias = torch.randn(N, device="cuda"); o16 = torch.empty(M, N, device="cuda", dtype=torch.half)
    line = f"{name:14s} {M:6d} {N:6d} {K:6d}"
    outs = []
    for v in variants:
        fn = lambda: ok(L.gdf_op_gemm_dit(P(A), K, P(W), P(bias), 0, None, 0, 0, 1, 0, 1, None, 0, None, 0, P(o16), N, None, 0, M, N, K, v, stream()), L)
        ms = t(fn); outs.append(o16.clone())
        line += f"  v{v}: {ms:7.4f} ms {2.0 * M * N * K / ms / 1e9:7.1f} TF"
    if len(outs) > 1: line += f"  maxdiff {float((outs[0].float() - outs[-1].float()).abs().max()):.3g}"
    print(line)
