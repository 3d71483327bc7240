#!/usr/bin/env python3
"""Experiment: N independent half-batch forwards in flight on N streams (natural de-phasing of MFMA-bound main loops and
HBM-bound epilogues / norms across CUs) vs one full-batch forward.   python tools/bench_dual.py [--steps 10]"""
import argparse, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "generic-diffusion-feature_amd")]
import torch
from components.native import NativeUNet, ARCH_CONFIGS
import bench as BB

ap = argparse.ArgumentParser(); ap.add_argument("--steps", type=int, default=10); ap.add_argument("--version", default="xl")
a = ap.parse_args()
cfg = ARCH_CONFIGS[a.version]; dev = torch.device("cuda:0"); lat = 128 if a.version == "xl" else 64
ids = BB.PRACTICAL[a.version]

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

def run(nstreams, B):
    nets = [NativeUNet(cfg, device=dev).init_synthetic(seed=0) for _ in range(nstreams)]
    ins = [inputs(B, 1 + i) for i in range(nstreams)]
    def step():
        for n, i in zip(nets, ins):
            n.forward_raw(*i, hook_ids=ids, shared_ctx=True)
    for _ in range(4): step()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(a.steps): step()
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    r = dict(streams=nstreams, batch_per_stream=B, images_per_s=round(nstreams * B * a.steps / dt, 2), ms_per_round=round(1e3 * dt / a.steps, 2))
    print(json.dumps(r), flush=True)
    del nets; torch.cuda.empty_cache()

ap2 = [(1, 16), (2, 8), (2, 16), (4, 4), (1, 32), (3, 8)] if a.version == "xl" else [(1, 32), (2, 32), (2, 16), (3, 32), (1, 64)]
for ns, B in ap2:
    run(ns, B)
