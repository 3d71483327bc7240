#!/usr/bin/env python3
"""Shape-level companion of tools/ab_epilogue_bound.sh: the GEMM / conv classes of the SDXL B = 16 step, the SD1.5 B = 32 step and the VAE on
RANDOM operands through the C ABI (gdf_op_gemm / gdf_op_conv3x3), so that the three builds (full / -DGDF_ABLATE_EPI=1 main loop only /
-DGDF_ABLATE_EPI=2 epilogue only) multiply the same data (inside a plan the ablated builds feed on unwritten buffers: low-toggle operands,
higher clocks).  Prints us per launch; run once per build."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tools"))
from ops_binding import P, lib, ok, stream
from bench_ops import timeit
L = lib(); dev = "cuda"
g = torch.Generator(device=dev).manual_seed(0)
R = lambda *s: torch.randn(*s, device=dev, generator=g)
print("# dense GEMMs: name M N K epilogue -> us")
for (name, M, N, K, epi) in [("sdxl qkv L2", 16384, 3840, 1280, "o16"), ("sdxl qkv L1", 65536, 1920, 640, "o16"),
                             ("sdxl attn_out L2", 16384, 1280, 1280, "res32"), ("sdxl attn_out L1", 65536, 640, 640, "res32"),
                             ("sdxl ff_out L2", 16384, 1280, 5120, "res32"), ("sdxl ff_out L1", 65536, 640, 2560, "res32"),
                             ("sdxl geglu L2", 16384, 10240, 1280, "geglu"), ("sdxl geglu L1", 65536, 5120, 640, "geglu"),
                             ("sd15 attn_out L0 B32", 131072, 320, 320, "res32"), ("sd15 ff_out L0 B32", 131072, 320, 1280, "res32"),
                             ("sd15 geglu L0 B32", 131072, 2560, 320, "geglu")]:
    A = R(M, K).half(); W = (R(N, K) * K ** -0.5).half(); bias = R(N)
    No = N // 2 if epi == "geglu" else N
    o16 = torch.empty(M, No, device=dev, dtype=torch.half); o32 = torch.empty(M, No, device=dev) if epi == "res32" else None
    res = R(M, No) if epi == "res32" else None
    fn = lambda: ok(L.gdf_op_gemm(P(A), K, P(W), P(bias), P(res) if res is not None else None, None, No, P(o16), No, P(o32) if o32 is not None else None, No,
                                  M, N, K, 1 if epi == "geglu" else 0, stream()), L)
    t = timeit(fn)
    print(f"{name:24s} {M:7d} {N:6d} {K:5d} {epi:6s} {t * 1e3:9.1f} us  {2.0 * M * N * K / t / 1e9:7.1f} TF", flush=True)
    del A, W, o16, o32, res
print("# 3x3 convs: name B H Cin Cout epilogue -> us")
for (name, B, H, Ci, Co, epi) in [("sdxl conv1 L0", 16, 128, 320, 320, "o16"), ("sdxl conv2 L0", 16, 128, 320, 320, "res32"),
                                  ("sdxl conv1 L2", 16, 32, 1280, 1280, "o16"), ("sdxl conv2 L2", 16, 32, 1280, 1280, "res32"),
                                  ("sdxl conv1 L1 cat", 16, 64, 1920, 640, "o16"), ("sdxl conv2 L1", 16, 64, 640, 640, "res32"),
                                  ("vae conv1 128@1024", 4, 1024, 128, 128, "o16"), ("vae conv2 128@1024", 4, 1024, 128, 128, "res32"),
                                  ("vae conv1 256@512", 4, 512, 256, 256, "o16"), ("vae conv2 256@512", 4, 512, 256, 256, "res32"),
                                  ("vae conv2 512@256", 4, 256, 512, 512, "res32")]:
    x = R(B, H, H, Ci).half(); w = (R(Co, 9 * Ci) * (9 * Ci) ** -0.5).half(); bias = R(Co)
    o16 = torch.empty(B, H, H, Co, device=dev, dtype=torch.half)
    res = R(B, H, H, Co) if epi == "res32" else None; o32 = torch.empty(B, H, H, Co, device=dev) if epi == "res32" else None
    fn = lambda: ok(L.gdf_op_conv3x3(P(x), Ci, B, H, H, Ci, P(w), Co, P(bias), None, 1, 0, P(res) if res is not None else None, None, P(o16),
                                     P(o32) if o32 is not None else None, 0, stream()), L)
    t = timeit(fn)
    print(f"{name:24s} {B:3d} {H:5d} {Ci:5d} {Co:5d} {epi:6s} {t * 1e3:9.1f} us  {2.0 * B * H * H * Co * 9 * Ci / t / 1e9:7.1f} TF", flush=True)
    del x, w, o16, o32, res
