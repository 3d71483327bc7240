#!/usr/bin/env python3
"""Experiment: two half-batch forwards side by side on DISJOINT CU partitions (CU-masked streams) vs one full-batch forward on the
whole chip.  A GEMM workgroup owns its CU, and all workgroups of a launch reach their HBM-bound epilogue together; on disjoint
partitions two chains drift out of phase and one chain's epilogues / norm passes meet the other chain's main loops.
    python tools/bench_cupart.py [--steps 10] [--census]"""
import argparse, json, os, sys, time, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "generic-diffusion-feature_amd")]
import torch
from components import native
from components.native import NativeUNet, ARCH_CONFIGS, make_cu_partition_streams
import bench as BB

ap = argparse.ArgumentParser(); ap.add_argument("--steps", type=int, default=10); ap.add_argument("--version", default="xl")
ap.add_argument("--census", action="store_true"); ap.add_argument("--layouts", default="interleave,block")
ap.add_argument("--batch", type=int, default=16)
a = ap.parse_args()
cfg = ARCH_CONFIGS[a.version]; dev = torch.device("cuda:0"); lat = 128 if a.version == "xl" else 64
ids = BB.PRACTICAL[a.version]
lib = native.load_library()

def census(stream, label):
    n = 2048
    out = torch.zeros(2 * n, dtype=torch.int32, device=dev)
    with torch.cuda.stream(stream):
        native._check(lib.gdf_cu_census(C.c_void_p(out.data_ptr()), n, 200, C.c_void_p(stream.cuda_stream)), "census")
    stream.synchronize()
    v = out.cpu().view(n, 2)
    xcc = (v[:, 0] & 0xf).tolist(); hw = v[:, 1].tolist()
    cus = {}
    for x, h in zip(xcc, hw):
        cu = (h >> 8) & 0xf; sh = (h >> 12) & 1; se = (h >> 13) & 0x7
        cus.setdefault(x, set()).add((se, sh, cu))
    print(label, "distinct CUs per XCC:", {k: len(s) for k, s in sorted(cus.items())}, "total", sum(len(s) for s in cus.values()), flush=True)
    return cus

def inputs(B, seed):
    g = torch.Generator(device=dev).manual_seed(seed)
    x = torch.randn(B, 4, lat, lat, generator=g, device=dev).half()
    ctx = torch.randn(1, 77, cfg["cross_attention_dim"], generator=g, device=dev).half().expand(B, -1, -1).contiguous()
    t = torch.full((B,), 100.0, device=dev); txt = tid = None
    if cfg["addition_embed_text_time"]:
        pooled = cfg["add_in_dim"] - 6 * cfg["addition_time_embed_dim"]
        txt = torch.randn(1, pooled, generator=g, device=dev).half().expand(B, -1).contiguous()
        tid = torch.tensor([[1024, 1024, 0, 0, 1024, 1024]], dtype=torch.float32, device=dev).repeat(B, 1)
    return x, t, ctx, txt, tid

def run(label, nets, B, steps):
    ins = [inputs(B, 1 + i) for i in range(len(nets))]
    user = [torch.cuda.Stream(device=dev) for _ in nets]      # one caller stream per chain: no cross-chain dependency through the caller
    torch.cuda.synchronize()
    def step():
        for n, i, u in zip(nets, ins, user):
            with torch.cuda.stream(u):
                n.forward_raw(*i, hook_ids=ids, shared_ctx=True)
    for _ in range(4): step()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(steps): step()
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    gs = [n._plans[next(iter(n._plans))].graph_stats() for n in nets]
    r = dict(label=label, chains=len(nets), batch_per_chain=B, images_per_s=round(len(nets) * B * steps / dt, 2),
             ms_per_round=round(1e3 * dt / steps, 2), graph_stats=gs)
    print(json.dumps(r), flush=True)
    return r

if a.census:
    census(torch.cuda.Stream(device=dev), "unmasked")
    for lay in a.layouts.split(","):
        ss, per = make_cu_partition_streams(dev, 2, lay)
        for k, s in enumerate(ss):
            census(s, f"{lay} part {k} ({per} CUs)")

base = NativeUNet(cfg, device=dev).init_synthetic(seed=0)
run("full chip", [base], a.batch, a.steps)
for lay in a.layouts.split(","):
    for parts in (2,):
        ss, per = make_cu_partition_streams(dev, parts, lay)
        nets = []
        for k in range(parts):
            n = NativeUNet(cfg, device=dev).init_synthetic(seed=0)
            n.cus = per; n.partition_stream = ss[k]
            nets.append(n)
        run(f"{parts} partitions x {per} CUs ({lay}), plans sized for the partition", nets, a.batch // parts, a.steps)
        run(f"{parts} partitions x {per} CUs ({lay}), full batch per chain", nets, a.batch, max(2, a.steps // 2))
        for n in nets:
            n.cus = 0; n._plans.clear()
        run(f"{parts} partitions x {per} CUs ({lay}), plans sized for the WHOLE chip", nets, a.batch // parts, a.steps)
        os.environ["GDF_HIP_GRAPH"] = "0"
        for n in nets:
            n.cus = per; n._plans.clear()
        run(f"{parts} partitions x {per} CUs ({lay}), eager launches", nets, a.batch // parts, a.steps)
        os.environ["GDF_HIP_GRAPH"] = "1"
        del nets
        torch.cuda.empty_cache()
run("full chip (again)", [base], a.batch, a.steps)
