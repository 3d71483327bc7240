#!/usr/bin/env python3
"""A few joint-attention launches at the Flux shape (24 heads x 128, 512 + 4096 tokens) for rocprofv3 --pmc runs."""
import ctypes, os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
from ops_binding import P, lib, ok, stream
L = lib(); dev = "cuda"
B, heads, T, S, D = int(sys.argv[1]) if len(sys.argv) > 1 else 2, 24, 512, 4096, 128
C = heads * D
buf = torch.randn(B * (T + S), 3 * C, device=dev).half()
o = torch.empty(B * (T + S), C, device=dev, dtype=torch.half)
ptr = lambda col: ctypes.c_void_p(buf.data_ptr() + col * 2)
for _ in range(3):
    ok(L.gdf_op_attention_joint(ptr(0), 3 * C, ptr(C), 3 * C, ptr(2 * C), 3 * C, P(o), C, B, heads, T, S, D, stream()), L)
torch.cuda.synchronize()
