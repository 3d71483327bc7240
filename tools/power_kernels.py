#!/usr/bin/env python3
"""Per-kernel power / clock: each kernel class of the SDXL step is looped for ~2 s while a thread samples the card's hwmon power and
shader clock.  Tells which kernels run at the 1400-W cap (clock-throttled: a cycle saved there comes back partly as lower clock) and
which do not.      python tools/power_kernels.py"""
import os, sys, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tools")]
import torch
from ops_binding import P, lib, ok, stream
from power_probe import find_nodes, Sampler, our_bus_id
L = lib(); dev = "cuda"
nodes = find_nodes(our_bus_id()); print('cards', len(nodes), our_bus_id(), flush=True)
flat = {os.path.basename(os.path.dirname(os.path.dirname(os.path.dirname(h)))) + ":" + k: p for h, d in nodes.items() for k, p in d.items()
        if k in ("power1_input", "freq1_input")}

def measure(name, fn, flops=0.0, bytes_=0.0, secs=2.0):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); 
    for _ in range(5): fn()
    e1.record(); torch.cuda.synchronize()
    per = e0.elapsed_time(e1) / 5
    n = max(10, int(secs * 1e3 / per))
    smp = Sampler(flat, dt=0.02); smp.start()
    t0 = time.perf_counter(); e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize(); t1 = time.perf_counter()
    smp.stop = True; smp.join()
    ms = e0.elapsed_time(e1) / n
    rows = [r for (ts, r) in smp.rows if t0 + 0.6 <= ts <= t1]
    best = max((k for k in flat if k.endswith("power1_input")), key=lambda k: sum(r[k] or 0 for r in rows))
    fk = best.replace("power1_input", "freq1_input")
    pw = sum(r[best] for r in rows) / max(1, len(rows)) / 1e6; fq = sum(r[fk] for r in rows) / max(1, len(rows)) / 1e6
    print(json.dumps(dict(kernel=name, us=round(ms * 1e3, 1), tflops=round(flops / ms / 1e9, 1) if flops else None,
                          tbps=round(bytes_ / ms / 1e9, 2) if bytes_ else None, watts=round(pw), mhz=round(fq), samples=len(rows))), flush=True)

def gemm(name, M, N, K, mode=""):
    A = torch.randn(M, K, device=dev).half(); W = (torch.randn(N, K, device=dev) * K ** -0.5).half(); bias = torch.randn(N, device=dev)
    No = N // 2 if mode == "geglu" else N
    o16 = torch.empty(M, No, device=dev, dtype=torch.half)
    o32 = torch.empty(M, No, device=dev) if mode == "res" else None
    res = torch.randn(M, No, device=dev) if mode == "res" else None
    flags = 1 if mode == "geglu" else 0
    measure(name, lambda: ok(L.gdf_op_gemm(P(A), K, P(W), P(bias), P(res), None, No, None if mode == "res" else P(o16), No, P(o32), No, M, N, K, flags, stream()), L),
            2.0 * M * N * K)

def conv(name, B, H, W, Ci, Co):
    x = torch.randn(B, H, W, Ci, device=dev).half(); w = (torch.randn(Co, 9 * Ci, device=dev) * (9 * Ci) ** -0.5).half(); bias = torch.randn(Co, device=dev)
    o16 = torch.empty(B, H, W, Co, device=dev, dtype=torch.half)
    measure(name, lambda: ok(L.gdf_op_conv3x3(P(x), Ci, B, H, W, Ci, P(w), Co, P(bias), None, 1, 0, None, None, P(o16), None, 0, stream()), L),
            2.0 * B * H * W * Co * 9 * Ci)

def attn(name, B, heads, S, Sk, D):
    q = torch.randn(B * S, heads * D, device=dev).half(); k = torch.randn(B * Sk, heads * D, device=dev).half(); v = torch.randn(B * Sk, heads * D, device=dev).half()
    o = torch.empty_like(q)
    measure(name, lambda: ok(L.gdf_op_attention(P(q), heads * D, P(k), heads * D, P(v), heads * D, P(o), heads * D, B, heads, S, Sk, D, None, stream()), L),
            4.0 * B * heads * S * Sk * D)

def ln(name, R, C):
    x = torch.randn(R, C, device=dev); g = torch.ones(C, device=dev); b = torch.zeros(C, device=dev); y = torch.empty(R, C, device=dev, dtype=torch.half)
    measure(name, lambda: ok(L.gdf_op_layernorm(None, P(x), C, R, C, 1e-5, P(g), P(b), P(y), stream()), L), 0.0, R * C * 6.0)

def zeros_gemm(name, M, N, K):
    A = torch.zeros(M, K, device=dev).half(); W = torch.zeros(N, K, device=dev).half(); bias = torch.zeros(N, device=dev)
    o16 = torch.empty(M, N, device=dev, dtype=torch.half)
    measure(name, lambda: ok(L.gdf_op_gemm(P(A), K, P(W), P(bias), None, None, N, P(o16), N, None, N, M, N, K, 0, stream()), L), 2.0 * M * N * K)

time.sleep(1.0)
gemm("gemm 8192^3", 8192, 8192, 8192)
zeros_gemm("gemm 8192^3 zero operands", 8192, 8192, 8192)
gemm("ff_out 16384x1280x5120 res32", 16384, 1280, 5120, "res")
gemm("qkv 16384x3840x1280", 16384, 3840, 1280)
gemm("geglu 16384x10240x1280", 16384, 10240, 1280, "geglu")
gemm("attn_out 16384x1280x1280 res32", 16384, 1280, 1280, "res")
conv("conv 640@64 B16", 16, 64, 64, 640, 640)
conv("conv 320@128 B16", 16, 128, 128, 320, 320)
attn("attn S4096 h10 D64", 16, 10, 4096, 4096, 64)
attn("attn S1024 h20 D64", 16, 20, 1024, 1024, 64)
attn("cross S1024x77 h20", 16, 20, 1024, 77, 64)
ln("layernorm 65536x1280", 65536, 1280)
ln("layernorm 16384x1280", 16384, 1280)
