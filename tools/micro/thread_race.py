#!/usr/bin/env python3
"""Diagnostics for concurrent forwards from two host threads (one model + plan per thread): each thread first computes a baseline of EVERY hook on fixed
inputs while the other thread idles, then both run the same forwards concurrently; the first hook (execution order) whose bits differ from the thread's
own baseline is reported.  GDF_HIP_GRAPH=0/1, RACE_MODE=raw (NativeUNet.forward_raw on device tensors) | extract (FeatureExtractor.extract)."""
import os
import sys
import threading

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "generic-diffusion-feature_amd"))
os.environ.setdefault("GDF_SYNTHETIC_WEIGHTS", "1")
import torch  # noqa: E402

from components.native import ARCH_CONFIGS, NativeUNet  # noqa: E402

N_ITER = int(os.environ.get("RACE_ITERS", "60"))
B = int(os.environ.get("RACE_B", "4"))
LAT = int(os.environ.get("RACE_LAT", "32"))
dev = torch.device("cuda:0")
torch.cuda.set_device(dev)
models, inputs, base, ids = [], [], [], None
VER = os.environ.get("RACE_VER", "1-5")
CFG = ARCH_CONFIGS[VER]
XL = bool(CFG["addition_embed_text_time"])
EXTRA = {}
for i in range(2):
    u = NativeUNet(CFG, device=dev)
    u.init_synthetic(seed=0)
    models.append(u)
    g = torch.Generator(device=dev).manual_seed(10 + i)
    inputs.append((torch.randn(B, 4, LAT, LAT, generator=g, device=dev).half(),
                   torch.randn(1, 77, CFG["cross_attention_dim"], generator=g, device=dev).half().expand(B, -1, -1).contiguous()))
    if XL:
        pooled = CFG["add_in_dim"] - 6 * CFG["addition_time_embed_dim"]
        EXTRA[i] = dict(text_embeds=torch.randn(1, pooled, generator=g, device=dev).half().expand(B, -1).contiguous(),
                        time_ids=torch.tensor([[8.0 * LAT, 8.0 * LAT, 0, 0, 8.0 * LAT, 8.0 * LAT]], device=dev).repeat(B, 1))
    else:
        EXTRA[i] = {}
ids = [h for h in models[0].hook_names() if os.environ.get("RACE_MAPS", "0") == "1" or not h.endswith("-map")]
if os.environ.get("RACE_FEW", "0") == "1":
    ids = ids[::9]
for i in range(2):
    x, ctx = inputs[i]
    for _ in range(3):
        n, h = models[i].forward_raw(x, 100.0, ctx, hook_ids=ids, shared_ctx=True, **EXTRA[i])
        torch.cuda.synchronize()
    base.append(({k: v.clone() for k, v in h.items()}, n.clone()))
    n2, h2 = models[i].forward_raw(x, 100.0, ctx, hook_ids=ids, shared_ctx=True, **EXTRA[i])
    torch.cuda.synchronize()
    assert all(torch.equal(h2[k], base[i][0][k]) for k in ids), "not even deterministic single-threaded"
# ---- where do the two plans' buffers live, and does anything write past a workspace? (RACE_GUARD=1: re-home every workspace inside a guarded buffer) ----
def plan_ranges(u):
    out = []
    for key, pl in u._plans.items():
        out.append(("workspace", pl.workspace.data_ptr(), pl.workspace.data_ptr() + pl.ws_bytes))
        for hs in pl.sets:
            out.append(("hookset", hs.lo, hs.hi))
        for nme, b in pl.staged.items():
            out.append(("staged:" + nme, b.data_ptr(), b.data_ptr() + b.numel() * b.element_size()))
    return out


R = [plan_ranges(models[0]), plan_ranges(models[1])]
for a in R[0]:
    for b in R[1]:
        if a[1] < b[2] and b[1] < a[2]:
            print(f"OVERLAP: thread0 {a[0]} [{a[1]:#x},{a[2]:#x}) with thread1 {b[0]} [{b[1]:#x},{b[2]:#x})")
for i in range(2):
    print(f"thread {i} buffers: " + ", ".join(f"{n} {lo:#x}+{(hi - lo) / 2 ** 20:.1f}MB" for n, lo, hi in R[i]))
GUARD = 32 << 20
guards = []
if os.environ.get("RACE_GUARD", "0") == "1":
    for u in models:
        for key, pl in u._plans.items():
            big = torch.full((pl.ws_bytes + 2 * GUARD,), 0xAB, dtype=torch.uint8, device=dev)
            pl.workspace = big[GUARD:GUARD + pl.ws_bytes]
            pl._guard = big
            guards.append((big, pl.ws_bytes))
    torch.cuda.synchronize()
print(f"baselines done: {len(ids)} hooks, B={B}, latent {LAT}, graph={os.environ.get('GDF_HIP_GRAPH', '1')}", flush=True)
report = [[], []]
bar = threading.Barrier(2)


def work(i):
    torch.cuda.set_device(dev)
    x, ctx = inputs[i]
    bar.wait()
    for it in range(N_ITER):
        n, h = models[i].forward_raw(x, 100.0, ctx, hook_ids=ids, shared_ctx=True, **EXTRA[i])
        if os.environ.get("RACE_SYNC", "device") == "device":
            torch.cuda.synchronize()
        else:
            torch.cuda.current_stream().synchronize()
        bad = [k for k in ids if not torch.equal(h[k], base[i][0][k])]
        if bad:
            k = bad[0]
            a, b = h[k].float(), base[i][0][k].float()
            rows = (a != b).reshape(a.shape[0], -1).any(1).tolist()
            d = (a != b)
            ext = []
            for ax in range(d.dim()):
                other = [x for x in range(d.dim()) if x != ax]
                idx = d.any(dim=other).nonzero().flatten() if other else d.nonzero().flatten()
                ext.append(f"{int(idx.min())}..{int(idx.max())}({idx.numel()}/{d.shape[ax]})")
            report[i].append((it, k, len(bad), float((a - b).norm() / b.norm()), rows, f"shape {tuple(a.shape)} strides {tuple(h[k].stride())} extents {ext} max|d| {float((a-b).abs().max()):.3g} max|ref| {float(b.abs().max()):.3g}"))
        del n, h


ths = [threading.Thread(target=work, args=(i,)) for i in range(2)]
[t.start() for t in ths]
[t.join() for t in ths]
torch.cuda.synchronize()
for gi, (big, n) in enumerate(guards):
    lo_bad = int((big[:GUARD] != 0xAB).sum()); hi_bad = int((big[GUARD + n:] != 0xAB).sum())
    print(f"guard {gi}: {lo_bad} bytes changed BELOW the workspace, {hi_bad} ABOVE it")
for i in range(2):
    print(f"thread {i}: {len(report[i])} of {N_ITER} forwards differ from the thread's own baseline")
    for r in report[i][:6]:
        print(f"   iter {r[0]}: first differing hook {r[1]} ({r[2]} hooks differ), rel {r[3]:.2e}, samples {r[4]}  {r[5]}")
