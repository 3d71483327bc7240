#!/usr/bin/env python3
"""One long-K GEMM (256x320 variant) for PMC collection."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
from ops_binding import P, lib, ok, stream
L = lib(); dev = "cuda"
M, N, K = 16384, 1280, 10240
var = int(sys.argv[1]) if len(sys.argv) > 1 else 320
A = torch.randn(M, K, device=dev).half(); W = (torch.randn(N, K, device=dev) * K ** -0.5).half()
o16 = torch.empty(M, N, device=dev, dtype=torch.half)
for _ in range(3):
    ok(L.gdf_op_gemm(P(A), K, P(W), None, None, None, N, P(o16), N, None, N, M, N, K, var << 8, stream()), L)
torch.cuda.synchronize()
