import os, sys, torch
ROOT = "/root/repo"
sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tools"))
from ops_binding import P, lib, ok, stream
from bench_ops import timeit
L = lib(); dev = "cuda"
g = torch.Generator(device=dev).manual_seed(0)
R = lambda *s: torch.randn(*s, device=dev, generator=g)
for (name, M, N, K, epi, var) in [("sd15 attn_out L0 B32", 131072, 320, 320, "res32", 0), ("sd15 ff_out L0 B32", 131072, 320, 1280, "res32", 0), ("sd15 proj L1 B32", 32768, 640, 640, "res32", 160),
                                  ("sdxl attn_out L1 (tile 160)", 65536, 640, 640, "res32", 160), ("sdxl attn_out L2 (tile 160)", 16384, 1280, 1280, "res32", 160),
                                  ("sdxl ff_out L1 (tile 160)", 65536, 640, 2560, "res32", 160), ("sdxl qkv L1 (tile 160)", 65536, 1920, 640, "o16", 160)]:
    A = R(M, K).half(); W = (R(N, K) * K ** -0.5).half(); bias = R(N)
    o16 = torch.empty(M, N, device=dev, dtype=torch.half); o32 = torch.empty(M, N, device=dev) if epi == "res32" else None
    res = R(M, N) if epi == "res32" else None
    fn = lambda: ok(L.gdf_op_gemm(P(A), K, P(W), P(bias), P(res) if res is not None else None, None, N, P(o16), N, P(o32) if o32 is not None else None, N, M, N, K, var << 8, stream()), L)
    t = timeit(fn)
    print(f"{name:30s} {M:7d} {N:5d} {K:5d} {t * 1e3:9.1f} us  {2.0 * M * N * K / t / 1e9:7.1f} TF", flush=True)
