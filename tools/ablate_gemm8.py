#!/usr/bin/env python3
"""A/B of the two-group 256x256 MMDiT / GEGLU GEMM main loop: shipped schedule (two 32-MFMA phases per K-tile) vs the round-1
four-phase schedule (tools/ablate_gemm.sh builds tools/micro/build/libgdf_phases4.so with -DGDF_PHASES4); results must be
bit-identical.   python tools/ablate_gemm8.py [rounds]"""
import ctypes as C, os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
vp, ci = C.c_void_p, C.c_int
libs = {"shipped": os.path.join(ROOT, "generic-diffusion-feature_amd", "libgdf.so"), "phases4": os.path.join(ROOT, "tools/micro/build/libgdf_phases4.so")}
shapes = [(36864, 9216, 3072), (36864, 3072, 15360), (36864, 12288, 3072), (8192, 8192, 8192), (16384, 10240, 1280)]
dev = "cuda"
L = {}
for k, p in libs.items():
    l = C.CDLL(p)
    l.gdf_op_gemm_dit.restype = ci
    l.gdf_op_gemm_dit.argtypes = [vp, ci, vp, vp, ci, vp, ci, ci, ci, ci, ci, vp, ci, vp, ci, vp, ci, vp, ci, ci, ci, ci, ci, vp]
    L[k] = l
keep = {}
for rnd in range(int(sys.argv[1]) if len(sys.argv) > 1 else 2):
    for k, l in L.items():
        row = []
        for M, N, K in shapes:
            if (M, N, K) not in keep:
                keep[(M, N, K)] = (torch.randn(M, K, device=dev).half(), (torch.randn(N, K, device=dev) * K ** -0.5).half(), torch.randn(N, device=dev))
            A, W, b = keep[(M, N, K)]
            o = torch.empty(M, N, device=dev, dtype=torch.half)
            s = vp(torch.cuda.current_stream().cuda_stream)
            fn = lambda: l.gdf_op_gemm_dit(vp(A.data_ptr()), K, vp(W.data_ptr()), vp(b.data_ptr()), 0, None, 0, 0, 1, 0, 1, None, 0, None, 0, vp(o.data_ptr()), N, None, 0, M, N, K, 8256, s)
            for _ in range(3): assert fn() == 0
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10): fn()
            e1.record(); torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / 10
            tag = ""
            if k == "shipped": keep[("o", M, N, K)] = o.clone()
            elif ("o", M, N, K) in keep: tag = " =" if torch.equal(o, keep[("o", M, N, K)]) else " DIFFERS"
            row.append(f"{M}x{N}x{K}: {ms:7.4f} ms {2.0 * M * N * K / ms / 1e9:7.1f}{tag}")
        print(f"{k:8s} " + "   ".join(row), flush=True)
