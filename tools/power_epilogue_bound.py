#!/usr/bin/env python3
"""Power / clock of a GEMM class under the three builds of tools/ab_epilogue_bound.sh (full, main loop only, epilogue only): each shape is
looped for ~2 s on random operands while the card's hwmon power and shader clock are sampled.  Joules per launch = watts x time: a schedule
that overlaps the epilogue with the main loop cannot finish before (E_main + E_epi - static x t) / (cap - static).  Run once per build."""
import os, sys, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tools")]
import torch
from ops_binding import P, lib, ok, stream
from power_probe import find_nodes, Sampler, our_bus_id
L = lib(); dev = "cuda"
nodes = find_nodes(our_bus_id())
flat = {os.path.basename(os.path.dirname(os.path.dirname(os.path.dirname(h)))) + ":" + k: p for h, d in nodes.items() for k, p in d.items()
        if k in ("power1_input", "freq1_input")}
g = torch.Generator(device=dev).manual_seed(0)
R = lambda *s: torch.randn(*s, device=dev, generator=g)


def measure(name, fn, flops, secs=2.5):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5): fn()
    e1.record(); torch.cuda.synchronize()
    n = max(10, int(secs * 1e3 / (e0.elapsed_time(e1) / 5)))
    smp = Sampler(flat, dt=0.02); smp.start()
    t0 = time.perf_counter(); e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize(); t1 = time.perf_counter()
    smp.stop = True; smp.join()
    ms = e0.elapsed_time(e1) / n
    rows = [r for (ts, r) in smp.rows if t0 + 0.7 <= ts <= t1]
    best = max((k for k in flat if k.endswith("power1_input")), key=lambda k: sum(r[k] or 0 for r in rows))
    fk = best.replace("power1_input", "freq1_input")
    pw = sum(r[best] for r in rows) / max(1, len(rows)) / 1e6; fq = sum(r[fk] for r in rows) / max(1, len(rows)) / 1e6
    print(json.dumps(dict(op=name, us=round(ms * 1e3, 1), tflops=round(flops / ms / 1e9, 1), watts=round(pw), mhz=round(fq),
                          millijoule_per_launch=round(pw * ms, 1), samples=len(rows))), flush=True)


for (name, M, N, K, epi) in [("sdxl qkv L2 (fp16 store)", 16384, 3840, 1280, "o16"), ("sdxl ff_out L2 (fp32 residual)", 16384, 1280, 5120, "res32"),
                             ("sdxl attn_out L1 (fp32 residual)", 65536, 640, 640, "res32"), ("sdxl geglu L2", 16384, 10240, 1280, "geglu")]:
    A = R(M, K).half(); W = (R(N, K) * K ** -0.5).half(); bias = R(N)
    No = N // 2 if epi == "geglu" else N
    o16 = torch.empty(M, No, device=dev, dtype=torch.half); o32 = torch.empty(M, No, device=dev) if epi == "res32" else None
    res = R(M, No) if epi == "res32" else None
    measure(name, lambda: ok(L.gdf_op_gemm(P(A), K, P(W), P(bias), P(res) if res is not None else None, None, No, P(o16), No,
                                           P(o32) if o32 is not None else None, No, M, N, K, 1 if epi == "geglu" else 0, stream()), L), 2.0 * M * N * K)
    del A, W, o16, o32, res
B, H, Ci, Co = 4, 1024, 128, 128
x = R(B, H, H, Ci).half(); w = (R(Co, 9 * Ci) * (9 * Ci) ** -0.5).half(); bias = R(Co)
o16 = torch.empty(B, H, H, Co, device=dev, dtype=torch.half); res = R(B, H, H, Co); o32 = torch.empty(B, H, H, Co, device=dev)
measure("vae conv2 128@1024 (fp32 residual)", lambda: ok(L.gdf_op_conv3x3(P(x), Ci, B, H, H, Ci, P(w), Co, P(bias), None, 1, 0, P(res), None, P(o16), P(o32), 0, stream()), L),
        2.0 * B * H * H * Co * 9 * Ci)
