#!/usr/bin/env python3
"""A few '-map' attention launches at the SD1.5 level-0 shape (8 heads x 40, 4096 tokens) for rocprofv3 --pmc / timing."""
import ctypes, os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
from ops_binding import P, lib, ok, stream
L = lib()
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
h = int(sys.argv[2]) if len(sys.argv) > 2 else 8
D = int(sys.argv[3]) if len(sys.argv) > 3 else 40
S = int(sys.argv[4]) if len(sys.argv) > 4 else 4096
C = h * D
qkv = torch.randn(B * S, 3 * C, device="cuda").half(); o = torch.empty(B * S, C, device="cuda", dtype=torch.half)
m = torch.empty(B, h, S, S, device="cuda", dtype=torch.half)
pk = ctypes.c_void_p(qkv.data_ptr() + C * 2); pv = ctypes.c_void_p(qkv.data_ptr() + 2 * C * 2)
f = lambda: ok(L.gdf_op_attention(P(qkv), 3 * C, pk, 3 * C, pv, 3 * C, P(o), C, B, h, S, S, D, P(m), stream()), L)
for _ in range(2): f()
torch.cuda.synchronize(); e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True); e0.record()
for _ in range(3): f()
e1.record(); torch.cuda.synchronize(); ms = e0.elapsed_time(e1) / 3
print(f"map attention B={B} h={h} D={D} S={S}: {ms:.3f} ms, map write {m.numel() * 2 / ms / 1e9:.2f} TB/s, {4.0 * B * h * S * S * D / ms / 1e9:.0f} TFLOP/s (single-pass count)")
