#!/usr/bin/env python3
"""Cost of keeping operand classes split (components/native.py SPLIT_CLASSES): images/s of the headline step per class mask.
    python tools/bench_split.py [--version xl] [--batch 16] [--steps 10] [--masks 0,stream,selective,precise]"""
import argparse, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "generic-diffusion-feature_amd")]
import torch
from components.native import NativeUNet, ARCH_CONFIGS, split_mask
import bench as BB
ap = argparse.ArgumentParser(); ap.add_argument("--version", default="xl"); ap.add_argument("--batch", type=int, default=0)
ap.add_argument("--steps", type=int, default=10)
ap.add_argument("--masks", default="0;stream;stream,gnv;stream,gnv,out;selective;selective,ff_inner;selective,ln_attn;stream,gnv,out,res;precise")
ap.add_argument("--hooks", default="practical")
a = ap.parse_args()
cfg = ARCH_CONFIGS[a.version]; dev = torch.device("cuda:0"); lat = 128 if a.version == "xl" else 64
B = a.batch or (16 if a.version == "xl" else 32)
g = torch.Generator(device=dev).manual_seed(1)
x = torch.randn(B, 4, lat, lat, generator=g, device=dev).half()
ctx = torch.randn(1, 77, cfg["cross_attention_dim"], generator=g, device=dev).half().expand(B, -1, -1).contiguous()
t = torch.full((B,), 100.0, device=dev); txt = tid = None
if cfg["addition_embed_text_time"]:
    pooled = cfg["add_in_dim"] - 6 * cfg["addition_time_embed_dim"]
    txt = torch.randn(1, pooled, generator=g, device=dev).half().expand(B, -1).contiguous()
    tid = torch.tensor([[1024, 1024, 0, 0, 1024, 1024]], dtype=torch.float32, device=dev).repeat(B, 1)
net = NativeUNet(cfg, device=dev).init_synthetic(seed=0)
ids = BB.PRACTICAL[a.version] if a.hooks == "practical" else [h for h in net.hook_names() if not h.endswith("-map")]
base = None
for spec in a.masks.split(";"):
    net.set_precise(spec); net._plans.clear(); torch.cuda.empty_cache()
    for _ in range(4): net.forward_raw(x, t, ctx, txt, tid, hook_ids=ids, shared_ctx=True)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(a.steps): net.forward_raw(x, t, ctx, txt, tid, hook_ids=ids, shared_ctx=True)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    ips = B * a.steps / dt
    base = base or ips
    print(json.dumps(dict(version=a.version, batch=B, hooks=len(ids), split=spec, mask=net.last_split, images_per_s=round(ips, 2), ms_per_step=round(1e3 * dt / a.steps, 2),
                          rel=round(ips / base, 3))), flush=True)
