#!/usr/bin/env python3
"""Where does a GEMM tile's fixed cost go?  Runs shapes on the time-stamped build (tools/trace_gemm.sh) and prints, per launch:
entry -> DMA issued -> K-tile 0 landed -> main loop done -> epilogue done (median / p90 over workgroups, us), the gap between a
workgroup's end and the start of the next one on the same CU, and the launch's span.   python tools/trace_gemm.py"""
import ctypes as C, os, sys, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
vp, ci = C.c_void_p, C.c_int
L = C.CDLL(os.path.join(ROOT, "tools/micro/build/libgdf_trace.so"))
L.gdf_op_gemm.restype = ci
L.gdf_op_gemm.argtypes = [vp, ci, vp, vp, vp, vp, ci, vp, ci, vp, ci, ci, ci, ci, ci, vp]
L.gdf_debug_trace.restype = ci
dev = "cuda"
def run(name, M, N, K, var, geglu=0, res=False):
    A = torch.randn(M, K, device=dev).half(); W = (torch.randn(N, K, device=dev) * K ** -0.5).half(); bias = torch.randn(N, device=dev)
    No = N // 2 if geglu else N
    o16 = torch.empty(M, No, device=dev, dtype=torch.half)
    o32 = torch.empty(M, No, device=dev) if res else None; r32 = torch.randn(M, No, device=dev) if res else None
    P = lambda t: vp(t.data_ptr()) if t is not None else None
    s = vp(torch.cuda.current_stream().cuda_stream)
    fn = lambda: L.gdf_op_gemm(P(A), K, P(W), P(bias), P(r32), None, No, P(o16), No, P(o32), No, M, N, K, (var << 8) | geglu, s)
    for _ in range(3): assert fn() == 0
    torch.cuda.synchronize()
    bm, bn = (256, 256) if var in (825, 826) else (256, 320)
    nwg = ((M + bm - 1) // bm) * ((N + bn - 1) // bn)
    buf = np.zeros(nwg * 8, dtype=np.uint64)
    assert L.gdf_debug_trace(buf.ctypes.data_as(C.POINTER(C.c_ulonglong)), nwg * 8) == 0
    t = buf.reshape(nwg, 8)
    ts = t[:, :5].astype(np.int64); base = ts[:, 0].min()
    us = (ts - base) / 100.0                                  # 100 MHz
    t6 = t[:, 6].astype(np.int64); hw = t6 & 0xffffffff; xcc = (t6 >> 32) & 0xf
    cu = ((xcc << 8) | (((hw >> 13) & 0x7) << 4) | ((hw >> 8) & 0xf))     # XCC, SE_ID (15:13), CU_ID (11:8)
    seg = np.diff(us, axis=1)
    q = lambda v: f"{np.median(v):6.2f} / {np.percentile(v, 90):6.2f}"
    gaps = []
    for c in np.unique(cu):
        idx = np.where(cu == c)[0]; idx = idx[np.argsort(us[idx, 0])]
        for a, b in zip(idx[:-1], idx[1:]): gaps.append(us[b, 0] - us[a, 4])
    print(f"{name} {M}x{N}x{K}: {nwg} workgroups on {len(np.unique(cu))} CUs, span {us[:, 4].max():7.1f} us | first start spread {np.percentile(us[:, 0], 10):.1f}")
    print(f"    setup+issue {q(seg[:, 0])}  tile0 wait {q(seg[:, 1])}  main loop {q(seg[:, 2])}  epilogue {q(seg[:, 3])}  total {q(us[:, 4] - us[:, 0])}"
          + (f"  end->next start on the CU {q(np.array(gaps))}" if gaps else ""), flush=True)
if len(sys.argv) > 1 and sys.argv[1] == "solo":
    # round 6: is a tile's epilogue time CU-local (instructions / latency) or a share of a chip-wide resource?  The same tiles with 4 ... 768
    # workgroups in flight: a CU-local epilogue takes the same time alone as in a full grid
    for M in (256, 1024, 4096, 16384):
        run("plain 932", M, 3840, 1280, 932)
        run("res32 932", M, 1280, 1280, 932, res=True)
        run("geglu 825", M, 10240, 1280, 825, 1)
    sys.exit(0)
for K in (64, 1280):
    run("geglu 825", 16384, 10240, K, 825, 1)
    run("plain 932", 16384, 3840, K, 932)
    run("plain 932", 16384, 1280, K, 932)
    run("res32 932", 16384, 1280, K, 932, res=True)
