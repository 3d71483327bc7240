#!/usr/bin/env python3
"""Per-op-name x kernel breakdown of one SDXL (or --version) step from the synchronising per-op pass.
    python tools/op_breakdown.py [--version xl] [--batch 16]"""
import argparse, os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "generic-diffusion-feature_amd"))
import bench
ap = argparse.ArgumentParser(); ap.add_argument("--version", default="xl"); ap.add_argument("--batch", type=int, default=16)
a = ap.parse_args()
from components.native import NativeUNet
cfg = bench._cfg(a.version); img = 1024 if a.version == "xl" else 512; lat = img // 8; B = a.batch; dev = "cuda:0"
unet = NativeUNet(cfg, device=dev); unet.init_synthetic(seed=0)
g = torch.Generator(device=dev).manual_seed(1)
x = torch.randn(B, 4, lat, lat, generator=g, device=dev).half()
ctx = torch.randn(1, 77, cfg["cross_attention_dim"], generator=g, device=dev).half().expand(B, -1, -1).contiguous()
t = torch.full((B,), 100.0, device=dev); txt = tid = None
if cfg["addition_embed_text_time"]:
    pooled = cfg["add_in_dim"] - 6 * cfg["addition_time_embed_dim"]
    txt = torch.randn(1, pooled, generator=g, device=dev).half().expand(B, -1).contiguous()
    tid = torch.tensor([[img, img, 0, 0, img, img]], dtype=torch.float32, device=dev).repeat(B, 1)
ids = bench.PRACTICAL[a.version]
for _ in range(2): unet.forward_raw(x, t, ctx, txt, tid, hook_ids=ids, shared_ctx=True)
acc = {}
for rep in range(3):
    _, _, prof = unet.forward_raw(x, t, ctx, txt, tid, hook_ids=ids, profile=True, shared_ctx=True)
    for name, ms, fl, lab in prof:
        d = acc.setdefault((name, lab), [0.0, 0.0, 0]); d[0] += ms / 3; d[1] += fl / 3; d[2] += 1
tot = sum(v[0] for v in acc.values())
print(f"total {tot:.2f} ms (sum of per-op synchronised times)")
for (name, lab), (ms, fl, n) in sorted(acc.items(), key=lambda kv: -kv[1][0]):
    print(f"{name:22s} {lab:42s} n={n // 3:4d} {ms:8.3f} ms {100 * ms / tot:5.1f} %  {fl / 1e9 / ms if ms > 0 else 0:7.1f} TF")
