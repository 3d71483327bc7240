#!/usr/bin/env python3
"""Race screen for the 8-phase GEMM main loops: every launch must reproduce the LDS-ring kernel bit for bit (same K order, same
fp32 accumulation), also while a bandwidth hog runs on a second stream and for K-tile counts 1..5, 20, 80.
    python tools/stress_gemm8.py [iterations]"""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
from ops_binding import P, lib, ok, stream
L = lib()
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 60
hog_a = torch.empty(256 << 20, device="cuda", dtype=torch.uint8); hog_b = torch.empty_like(hog_a)
side = torch.cuda.Stream()
bad = 0; total = 0
def run(desc, fn_ref, fn_new, out):
    global bad, total
    fn_ref(); torch.cuda.synchronize(); ref = out.clone()
    for i in range(iters):
        if i % 3 == 1:
            with torch.cuda.stream(side): hog_b.copy_(hog_a)
        out.zero_(); fn_new(); torch.cuda.synchronize()
        total += 1
        if not torch.equal(out, ref):
            bad += 1; print("MISMATCH", desc, i, float((out.float() - ref.float()).abs().max()))
    print(f"{desc:48s} ok" if bad == 0 else f"{desc} FAILED")
for M, N, K in [(4096, 1280, 64), (4096, 1280, 128), (2048, 3840, 192), (2048, 1280, 256), (4100, 1280, 320), (16384, 1280, 1280),
                (16384, 3840, 1280), (8192, 1280, 5120)]:
    A = torch.randn(M, K, device="cuda").half(); W = (torch.randn(N, K, device="cuda") * K ** -0.5).half(); bias = torch.randn(N, device="cuda")
    res = torch.randn(M, N, device="cuda"); o16 = torch.empty(M, N, device="cuda", dtype=torch.half); o32 = torch.empty(M, N, device="cuda")
    g = lambda var: (lambda: ok(L.gdf_op_gemm(P(A), K, P(W), P(bias), P(res), None, N, P(o16), N, P(o32), N, M, N, K, var << 8, stream()), L))
    run(f"dense 932 vs 320   {M}x{N}x{K} (o32)", g(320), g(932), o32)
for M, C in [(4096, 320), (16384, 1280), (8200, 640)]:
    K = C; N = 8 * C
    A = torch.randn(M, K, device="cuda").half(); W = (torch.randn(N, K, device="cuda") * K ** -0.5).half(); bias = torch.randn(N, device="cuda")
    o16 = torch.empty(M, N // 2, device="cuda", dtype=torch.half)
    g = lambda var: (lambda: ok(L.gdf_op_gemm(P(A), K, P(W), P(bias), None, None, N // 2, P(o16), N // 2, None, 0, M, N, K, (var << 8) | 1, stream()), L))
    run(f"geglu 825 vs 320   {M}x{N}x{K}", g(320), g(825), o16)
for M, N, K in [(4608, 3072, 64), (4608, 9216, 192), (9216, 3072, 3072), (4608, 1152, 1152), (36864, 3072, 3072)]:
    A = torch.randn(M, K, device="cuda").half(); W = (torch.randn(N, K, device="cuda") * K ** -0.5).half(); bias = torch.randn(N, device="cuda")
    o16 = torch.empty(M, N, device="cuda", dtype=torch.half)
    g = lambda var: (lambda: ok(L.gdf_op_gemm_dit(P(A), K, P(W), P(bias), 1, None, 0, 0, 1, 0, 1, None, 0, None, 0, P(o16), N, None, 0, M, N, K, var, stream()), L))
    run(f"dit 8256 vs 1256   {M}x{N}x{K} (gelu)", g(1256), g(8256), o16)
for B, H, Ci, Co in [(2, 64, 64, 320), (4, 64, 320, 640), (2, 32, 1280, 1280), (2, 64, 128, 256), (2, 48, 256, 512), (9, 64, 64, 1024)]:   # last: 144 x 4 = 576 tiles, persistent
    x = torch.randn(B, H, H, Ci, device="cuda").half(); w = (torch.randn(Co, 9 * Ci, device="cuda") * (9 * Ci) ** -0.5).half()
    bias = torch.randn(Co, device="cuda"); o16 = torch.empty(B, H, H, Co, device="cuda", dtype=torch.half)
    var_new, var_ref = (932, 320) if Co % 320 == 0 else (826, 256)
    g = lambda var: (lambda: ok(L.gdf_op_conv3x3(P(x), Ci, B, H, H, Ci, P(w), Co, P(bias), None, 1, 0, None, None, P(o16), None, var << 8, stream()), L))
    run(f"conv {var_new} vs {var_ref}   {B}x{H}x{H}x{Ci}->{Co}", g(var_ref), g(var_new), o16)
# attention-map kernel with the loader wave: repeated launches must agree bit for bit (its asm loads use hand-counted waits)
import ctypes
for B, h, S, D in [(2, 8, 1024, 40), (1, 4, 2048, 32)]:
    C = h * D
    qkv = torch.randn(B * S, 3 * C, device="cuda").half(); o = torch.empty(B * S, C, device="cuda", dtype=torch.half)
    m = torch.empty(B, h, S, S, device="cuda", dtype=torch.half)
    pk = ctypes.c_void_p(qkv.data_ptr() + C * 2); pv = ctypes.c_void_p(qkv.data_ptr() + 2 * C * 2)
    f = lambda: ok(L.gdf_op_attention(P(qkv), 3 * C, pk, 3 * C, pv, 3 * C, P(o), C, B, h, S, S, D, P(m), stream()), L)
    f(); torch.cuda.synchronize(); m0, o0 = m.clone(), o.clone()
    nb = 0
    for i in range(iters):
        if i % 3 == 1:
            with torch.cuda.stream(side): hog_b.copy_(hog_a)
        m.zero_(); o.zero_(); f(); torch.cuda.synchronize(); total += 1
        if not (torch.equal(m, m0) and torch.equal(o, o0)): nb += 1
    bad += nb
    print(f"{'attn map (loader wave) ' + str((B, h, S, D)):48s} {'ok' if nb == 0 else 'FAILED'}")
print(f"{total} launches, {bad} mismatches")
sys.exit(1 if bad else 0)
