#!/usr/bin/env python3
"""Quick attention timing: SDXL self-attention shapes (D=64), PixArt (D=72) and the Flux joint shape (D=128)."""
import ctypes, os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
from ops_binding import P, lib, ok, stream
L = lib()
def t(fn, it=10):
    for _ in range(3): fn()
    torch.cuda.synchronize(); e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True); e0.record()
    for _ in range(it): fn()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) / it
for name, B, h, S, D in (("sdxl_1024", 16, 20, 1024, 64), ("sdxl_4096", 16, 10, 4096, 64), ("pixart_4096_d72", 16, 16, 4096, 72), ("sd15_4096_d40", 8, 8, 4096, 40), ("sd15_4096_d40_b32", 32, 8, 4096, 40), ("sd15_1024_d80", 32, 8, 1024, 80), ("cross_4096x77_d40", 32, 8, 4096, 40), ("sd15_256_d160", 32, 8, 256, 160)):
    C = h * D
    qkv = torch.randn(B * S, 3 * C, device="cuda").half(); o = torch.empty(B * S, C, device="cuda", dtype=torch.half)
    pk = ctypes.c_void_p(qkv.data_ptr() + C * 2); pv = ctypes.c_void_p(qkv.data_ptr() + 2 * C * 2)
    Sk = 77 if name.startswith("cross") else S
    ms = t(lambda: ok(L.gdf_op_attention(P(qkv), 3 * C, pk, 3 * C, pv, 3 * C, P(o), C, B, h, S, Sk, D, None, stream()), L))
    # parity of the first sample / first two heads vs fp32 SDPA (catches a wrong fragment layout at once)
    q0 = qkv[:S, 0:2 * D].float().view(S, 2, D).transpose(0, 1); k0 = qkv[:Sk, C:C + 2 * D].float().view(Sk, 2, D).transpose(0, 1)
    v0 = qkv[:Sk, 2 * C:2 * C + 2 * D].float().view(Sk, 2, D).transpose(0, 1)
    ref = torch.softmax(q0 @ k0.transpose(1, 2) * D ** -0.5, -1) @ v0
    got = o[:S, :2 * D].float().view(S, 2, D).transpose(0, 1)
    err = float((got - ref).norm() / ref.norm())
    print(f"{name:18s} {ms:8.4f} ms {4.0 * B * h * S * Sk * D / ms / 1e9:8.1f} TFLOP/s   rel err vs fp32 {err:.2e}")
B, heads, T, S, D = 8, 24, 512, 4096, 128; C = heads * D
buf = torch.randn(B * (T + S), 3 * C, device="cuda").half(); o = torch.empty(B * (T + S), C, device="cuda", dtype=torch.half)
ptr = lambda col: ctypes.c_void_p(buf.data_ptr() + col * 2)
ms = t(lambda: ok(L.gdf_op_attention_joint(ptr(0), 3 * C, ptr(C), 3 * C, ptr(2 * C), 3 * C, P(o), C, B, heads, T, S, D, stream()), L))
print(f"{'flux_joint_d128':18s} {ms:8.4f} ms {4.0 * B * heads * (T + S) ** 2 * D / ms / 1e9:8.1f} TFLOP/s")
