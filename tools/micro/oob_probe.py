#!/usr/bin/env python3
"""Does a kernel write outside its output?  The op's output lives in the middle of a sentinel-filled allocation; every other input too; guards are checked after N launches."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import ctypes as C
from ops_binding import P, lib, ok, stream
L = lib()
torch.cuda.set_device(0)
G = 8 << 20   # guard halves on each side


def guarded(shape, dtype=torch.half, fill=None):
    n = 1
    for d in shape:
        n *= d
    big = torch.full((n + 2 * G,), 31337.0 if dtype != torch.uint8 else 0xAB, dtype=dtype, device="cuda")
    mid = big[G:G + n].view(shape)
    if fill is not None:
        mid.copy_(fill)
    return big, mid


def check(name, big, n):
    lo = int((big[:G] != 31337.0).sum()); hi = int((big[G + n:] != 31337.0).sum())
    print(f"  {name}: {lo} elements changed BELOW, {hi} ABOVE")


for (M, N, K, variant) in ((256, 9216, 3072, 0), (256, 3072, 3072, 0), (2304, 9216, 3072, 0), (4096, 3072, 3072, 8256), (200, 9216, 3072, 0)):
    g = torch.Generator().manual_seed(M + N)
    A = (torch.randn(M, K, generator=g)).half().cuda(); W = (torch.randn(N, K, generator=g) * K ** -0.5).half().cuda(); bias = torch.randn(N, generator=g).cuda()
    bigo, o16 = guarded((M, N))
    for _ in range(20):
        ok(L.gdf_op_gemm_dit(P(A), K, P(W), P(bias), 0, None, 0, 0, 1, 0, 1, None, 0, None, 0, P(o16), N, None, 0, M, N, K, variant, stream()), L)
    torch.cuda.synchronize()
    print(f"dit gemm {M}x{N}x{K} variant {variant}:")
    check("out16", bigo, M * N)
    ref = A.float() @ W.float().t() + bias
    print(f"  rel err {float((o16.float() - ref).norm() / ref.norm()):.2e}")
