#!/usr/bin/env python3
"""fp8-MX (e4m3, v_mfma_scale_f32_16x16x128_f8f6f4) vs bf16 MMDiT GEMM at the Flux shapes (include/gdf_ops.h gdf_op_gemm_mx / gdf_op_gemm_dit).
    python tools/bench_gemm_mx.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from ops_binding import P, lib, ok, stream
L = lib(); dev = "cuda"
def timeit(fn, iters=10, warm=3):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters
L.gdf_op_set_e16(2)
for name, M, N, K in (("qkv", 36864, 9216, 3072), ("proj_mlp", 36864, 12288, 3072), ("ff_out", 32768, 3072, 12288), ("proj_out", 36864, 3072, 15360), ("8192^3", 8192, 8192, 8192)):
    A = torch.randn(M, K, device=dev).bfloat16(); W = (torch.randn(N, K, device=dev) * K ** -0.5).bfloat16(); bias = torch.randn(N, device=dev)
    o16 = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    A8 = torch.empty(M, K, dtype=torch.uint8, device=dev); sa = torch.empty(M, device=dev)
    W8 = torch.empty(N, K, dtype=torch.uint8, device=dev); sw = torch.empty(N, device=dev)
    ok(L.gdf_op_quant_rows_fp8(P(W), K, N, K, 1, P(W8), K, P(sw), stream()), L)
    tq = timeit(lambda: ok(L.gdf_op_quant_rows_fp8(P(A), K, M, K, 1, P(A8), K, P(sa), stream()), L))
    tb = timeit(lambda: ok(L.gdf_op_gemm_dit(P(A), K, P(W), P(bias), 0, None, 0, 0, 1, 0, 1, None, 0, None, 0, P(o16), N, None, 0, M, N, K, 0, stream()), L))
    tm = timeit(lambda: ok(L.gdf_op_gemm_mx(P(A8), K, P(sa), P(W8), P(sw), P(bias), 0, None, 0, P(o16), N, None, 0, M, N, K, stream()), L))
    fl = 2.0 * M * N * K
    print(f"{name:9s} {M}x{N}x{K}: bf16 {tb:7.3f} ms {fl / tb / 1e9:7.1f} TF | fp8-mx {tm:7.3f} ms {fl / tm / 1e9:7.1f} TF ({tb / tm:.2f}x) | quantise A {tq:6.3f} ms "
          f"({M * K * 3 / tq / 1e9:.2f} TB/s) -> with it {fl / (tm + tq) / 1e9:7.1f} TF ({tb / (tm + tq):.2f}x)", flush=True)
