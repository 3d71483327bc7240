#!/usr/bin/env python3
"""Which hipBLASLt kernels does torch.matmul pick at our GEMM shapes (run under rocprofv3 --kernel-trace --stats)."""
import torch
for M, N, K in ((16384, 3840, 1280), (16384, 1280, 5120), (16384, 10240, 1280), (8192, 8192, 8192)):
    A = torch.randn(M, K, device="cuda").half(); W = torch.randn(N, K, device="cuda").half()
    o = torch.empty(M, N, device="cuda", dtype=torch.half)
    for _ in range(5): torch.matmul(A, W.t(), out=o)
torch.cuda.synchronize()
