import os, sys, torch
sys.path.insert(0, "tests")
from ops_binding import P, lib, ok, stream
L = lib()
def t(fn, it=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True); e0.record()
    for _ in range(it): fn()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) / it
for name, M, N, K in (("qkv", 16384, 3840, 1280), ("attn_out", 16384, 1280, 1280), ("ff_out", 16384, 1280, 5120)):
    A = torch.randn(M, K, device="cuda").half(); W = (torch.randn(N, K, device="cuda") * K ** -0.5).half()
    bias = torch.randn(N, device="cuda"); o16 = torch.empty(M, N, device="cuda", dtype=torch.half)
    o32 = torch.empty(M, N, device="cuda"); res = torch.randn(M, N, device="cuda")
    fl = 2.0 * M * N * K / 1e9
    a = t(lambda: ok(L.gdf_op_gemm(P(A), K, P(W), P(bias), None, None, N, None, N, None, N, M, N, K, 320 << 8, stream()), L))
    b = t(lambda: ok(L.gdf_op_gemm(P(A), K, P(W), P(bias), None, None, N, P(o16), N, None, N, M, N, K, 320 << 8, stream()), L))
    c = t(lambda: ok(L.gdf_op_gemm(P(A), K, P(W), P(bias), P(res), None, N, None, N, P(o32), N, M, N, K, 320 << 8, stream()), L))
    d = t(lambda: ok(L.gdf_op_gemm(P(A), K, P(W), P(bias), P(res), None, N, P(o16), N, P(o32), N, M, N, K, 320 << 8, stream()), L))
    print(f"{name:9s} no-store {a*1e3:7.1f} us ({fl/a:6.0f} TF) | out16 {b*1e3:7.1f} us ({fl/b:6.0f}) | res32+out32 {c*1e3:7.1f} us ({fl/c:6.0f}) | res32+out16+out32 {d*1e3:7.1f} us ({fl/d:6.0f})")
