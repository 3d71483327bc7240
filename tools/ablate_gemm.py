#!/usr/bin/env python3
"""Time the 256x320 9-phase GEMM main loop with parts compiled out (tools/ablate_gemm.sh builds the variants).
Prints ms and the TFLOP/s the full work would correspond to; results of the ablated builds are garbage by design."""
import ctypes as C, os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
vp, ci = C.c_void_p, C.c_int
names = {0: "full (2 phases, shipped)", 100: "4 phases (round 1)", 1: "no reads", 2: "no DMA", 3: "no reads, no DMA", 4: "no barriers", 8: "no MFMA", 9: "no MFMA, no reads", 10: "no MFMA, no DMA",
         11: "barriers only", 12: "no MFMA, no barriers"}
shapes = [(16384, 1280, 10240), (8192, 7680, 8192), (16384, 3840, 1280), (16384, 1280, 5120), (16384, 1280, 1280)]
dev = "cuda"
only = [int(x) for x in sys.argv[1:]]
keep = {}
_w = torch.randn(8192, 8192, device=dev).half()
for _ in range(150): _w @ _w
for a in (only or (0, 100, 1, 2, 3, 4, 8, 9, 10, 11, 12)):
    path = (os.path.join(ROOT, "generic-diffusion-feature_amd", "libgdf.so") if a == 0 else
            os.path.join(ROOT, "tools/micro/build", "libgdf_phases4.so" if a == 100 else f"libgdf_abl{a}.so"))
    L = C.CDLL(path)
    L.gdf_op_gemm.restype = ci
    L.gdf_op_gemm.argtypes = [vp, ci, vp, vp, vp, vp, ci, vp, ci, vp, ci, ci, ci, ci, ci, vp]
    row = []
    for M, N, K in shapes:
        A = torch.randn(M, K, device=dev).half(); W = (torch.randn(N, K, device=dev) * K ** -0.5).half()
        o = torch.empty(M, N, device=dev, dtype=torch.half)
        s = vp(torch.cuda.current_stream().cuda_stream)
        fn = lambda: L.gdf_op_gemm(vp(A.data_ptr()), K, vp(W.data_ptr()), None, None, None, N, vp(o.data_ptr()), N, None, N, M, N, K, 932 << 8, s)
        for _ in range(3): assert fn() == 0
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): fn()
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 10
        tag = ""
        if a == 0: keep[(M, N, K)] = (A, W, o.clone())
        elif a >= 100 and (M, N, K) in keep:
            A0, W0, o0 = keep[(M, N, K)]
            L.gdf_op_gemm(vp(A0.data_ptr()), K, vp(W0.data_ptr()), None, None, None, N, vp(o.data_ptr()), N, None, N, M, N, K, 932 << 8, s); torch.cuda.synchronize()
            tag = " bit-identical" if torch.equal(o, o0) else f" DIFFERS max {float((o.float() - o0.float()).abs().max()):.3g}"
        row.append(f"{M}x{N}x{K}: {ms:7.4f} ms {2.0 * M * N * K / ms / 1e9:7.1f}{tag}")
    print(f"{names[a]:22s} " + "   ".join(row), flush=True)

# ---- 3x3 convolutions (A_CONV3 on the same 256x320 kernel) ----
conv = [(16, 32, 32, 1280, 1280), (16, 64, 64, 640, 640), (16, 128, 128, 320, 320), (16, 32, 32, 2560, 1280)]
keepc = {}
for a in (only or (0, 100)):
    if a not in (0, 100): continue
    path = (os.path.join(ROOT, "generic-diffusion-feature_amd", "libgdf.so") if a == 0 else os.path.join(ROOT, "tools/micro/build", "libgdf_phases4.so"))
    L = C.CDLL(path)
    L.gdf_op_conv3x3.restype = ci
    L.gdf_op_conv3x3.argtypes = [vp, ci, ci, ci, ci, ci, vp, ci, vp, vp, ci, ci, vp, vp, vp, vp, ci, vp]
    L.gdf_op_relayout_conv3.restype = ci
    L.gdf_op_relayout_conv3.argtypes = [vp, vp, ci, ci, vp]
    row = []
    for B, H, Wd, Ci, Co in conv:
        key = (B, H, Wd, Ci, Co)
        if key not in keepc:
            x = torch.randn(B, H, Wd, Ci, device=dev).half(); w = (torch.randn(Co, Ci, 3, 3, device=dev) * (9 * Ci) ** -0.5).half()
            keepc[key] = (x, w, torch.randn(Co, device=dev))
        x, w, b = keepc[key]
        s = vp(torch.cuda.current_stream().cuda_stream)
        wl = torch.empty(Co, 9, Ci, device=dev, dtype=torch.half)
        assert L.gdf_op_relayout_conv3(vp(w.data_ptr()), vp(wl.data_ptr()), Co, Ci, s) == 0
        o = torch.empty(B, H, Wd, Co, device=dev, dtype=torch.half)
        fn = lambda: L.gdf_op_conv3x3(vp(x.data_ptr()), Ci, B, H, Wd, Ci, vp(wl.data_ptr()), Co, vp(b.data_ptr()), None, 1, 0, None, None, vp(o.data_ptr()), None, 0, s)
        for _ in range(3): assert fn() == 0
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): fn()
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 10
        tag = ""
        if a == 0: keepc[("o",) + key] = o.clone()
        elif ("o",) + key in keepc: tag = " =" if torch.equal(o, keepc[("o",) + key]) else " DIFFERS"
        row.append(f"conv {Ci}->{Co}@{H}: {ms:7.4f} ms {2.0 * B * H * Wd * Co * 9 * Ci / ms / 1e9:7.1f}{tag}")
    print(f"{names[a]:28s} " + "   ".join(row), flush=True)
