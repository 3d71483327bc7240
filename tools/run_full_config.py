#!/usr/bin/env python3
"""BASELINE config C2: SD1.5 512x512, batch 32, t=100, FULL layer set (197 ids incl. '-map' hooks), 1 GPU."""
import os, sys, time, json, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "generic-diffusion-feature_amd"))
from components.native import NativeUNet, ARCH_CONFIGS
ver = sys.argv[1] if len(sys.argv) > 1 else "1-5"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 32
maps = (sys.argv[3] != "nomaps") if len(sys.argv) > 3 else True
cfg = ARCH_CONFIGS[ver]; lat = 64 if ver == "1-5" else 128
dev = torch.device("cuda:0")
u = NativeUNet(cfg, device=dev).init_synthetic(0)
ids = [i for i in u.hook_names() if maps or "map" not in i]
g = torch.Generator(device=dev).manual_seed(1)
x = torch.randn(B, 4, lat, lat, generator=g, device=dev).half()
ctx = torch.randn(1, 77, cfg["cross_attention_dim"], generator=g, device=dev).half().expand(B, -1, -1).contiguous()
t = torch.full((B,), 100.0, device=dev)
txt = tid = None
if cfg["addition_embed_text_time"]:
    txt = torch.randn(1, 1280, generator=g, device=dev).half().expand(B, -1).contiguous()
    tid = torch.tensor([[lat * 8, lat * 8, 0, 0, lat * 8, lat * 8]], dtype=torch.float32, device=dev).repeat(B, 1)
for it in range(3):
    torch.cuda.synchronize(); t0 = time.time()
    _, hooks = u.forward_raw(x, t, ctx, txt, tid, hook_ids=ids, shared_ctx=True)
    torch.cuda.synchronize(); dt = time.time() - t0
    nbytes = sum(v.numel() * 2 for v in hooks.values())
    print(json.dumps(dict(version=ver, batch=B, hooks=len(hooks), maps=maps, hook_GB=round(nbytes / 1e9, 2), ms=round(dt * 1e3, 1),
                          images_per_s=round(B / dt, 1), hook_write_GBps=round(nbytes / dt / 1e9, 1))))
    del hooks
_, hooks, prof = u.forward_raw(x, t, ctx, txt, tid, hook_ids=ids, shared_ctx=True, profile=True)
rows = {}
for name, ms, fl, k in prof:
    r = rows.setdefault(name, [0.0, 0]); r[0] += ms; r[1] += 1
for name, r in sorted(rows.items(), key=lambda kv: -kv[1][0])[:8]:
    print(f"# {name:14s} n={r[1]:4d} {r[0]:9.2f} ms")
