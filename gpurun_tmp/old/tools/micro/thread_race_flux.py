#!/usr/bin/env python3
"""tools/micro/thread_race.py for the MMDiT: two host threads, one Flux transformer each (true widths, reduced depth 2 + 3 blocks, 1024 image + 128 text
tokens, B = 2), every hook of every concurrent forward compared with the thread's own single-threaded result.  RACE_DT = auto | bfloat16 | fp8-mx | bfloat16x2."""
import os, sys, threading
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "generic-diffusion-feature_amd")); sys.path.insert(0, ROOT)
import torch
from components.native import FLUX_CONFIGS, NativeFluxTransformer
from oracle import flux_ref as FR

N_ITER = int(os.environ.get("RACE_ITERS", "150"))
DT = os.environ.get("RACE_DT", "auto")
dev = torch.device("cuda:0"); torch.cuda.set_device(dev)
cfg = dict(FLUX_CONFIGS["flux"], num_layers=2, num_single_layers=3)
arch = dict(FR.ARCH_FLUX, num_layers=2, num_single_layers=3) if hasattr(FR, "ARCH_FLUX") else None
models, inputs, base = [], [], []
GRID, NTXT, B = 32, 128, 2
for i in range(2):
    net = NativeFluxTransformer(dict(cfg), device=dev, compute_dtype=DT)
    net.init_synthetic(seed=0)
    models.append(net)
    g = torch.Generator(device=dev).manual_seed(20 + i)
    dt16 = torch.bfloat16 if DT in ("bfloat16", "bfloat16x2", "fp8-mx") else torch.float16
    hs = torch.randn(B, GRID * GRID, 64, generator=g, device=dev).to(dt16)
    enc = torch.randn(B, NTXT, 4096, generator=g, device=dev).to(dt16)
    pooled = torch.randn(B, 768, generator=g, device=dev).to(dt16)
    img_ids = torch.zeros(GRID * GRID, 3, device=dev); img_ids[:, 1] = torch.arange(GRID, device=dev).repeat_interleave(GRID); img_ids[:, 2] = torch.arange(GRID, device=dev).repeat(GRID)
    txt_ids = torch.zeros(NTXT, 3, device=dev)
    inputs.append(dict(hidden_states=hs, encoder_hidden_states=enc, pooled_projections=pooled, timestep=torch.full((B,), 0.1, device=dev), img_ids=img_ids, txt_ids=txt_ids,
                       guidance=torch.full((B,), 1.0, device=dev)))
ids = [h for h in models[0].hook_names() if not h.endswith("-map")]


def fwd(i):
    I = inputs[i]
    return models[i].forward_raw(I["hidden_states"], I["encoder_hidden_states"], I["pooled_projections"], I["timestep"], I["img_ids"], I["txt_ids"], guidance=I["guidance"],
                                 hook_ids=ids, grid=(GRID, GRID))


for i in range(2):
    for _ in range(3):
        o, h = fwd(i); torch.cuda.synchronize()
    base.append({k: v.clone() for k, v in h.items()})
    o, h = fwd(i); torch.cuda.synchronize()
    assert all(torch.equal(h[k], base[i][k]) for k in ids)
print(f"baselines done: {len(ids)} hooks, mode {DT}, graph={os.environ.get('GDF_HIP_GRAPH', '1')}", flush=True)
bad = [0, 0]; first = [None, None]
bar = threading.Barrier(2)


def work(i):
    torch.cuda.set_device(dev)
    bar.wait()
    for it in range(N_ITER):
        o, h = fwd(i)
        torch.cuda.synchronize()
        b = [k for k in ids if not torch.equal(h[k], base[i][k])]
        if b:
            bad[i] += 1
            if first[i] is None:
                k = b[0]
                a_, b_ = h[k].float(), base[i][k].float()
                d = a_ != b_
                ext = []
                for ax in range(d.dim()):
                    other = [x for x in range(d.dim()) if x != ax]
                    idx = d.any(dim=other).nonzero().flatten()
                    ext.append(f"{int(idx.min())}..{int(idx.max())}({idx.numel()}/{d.shape[ax]})")
                first[i] = (it, k, len(b), f"rel {float((a_ - b_).norm() / b_.norm()):.2e} max|d| {float((a_ - b_).abs().max()):.3g} max|ref| {float(b_.abs().max()):.3g} shape {tuple(a_.shape)} extents {ext}")
        del o, h


ths = [threading.Thread(target=work, args=(i,)) for i in range(2)]
[t.start() for t in ths]; [t.join() for t in ths]
for i in range(2):
    print(f"thread {i}: {bad[i]} of {N_ITER} forwards differ from the thread's own baseline {first[i] or ''}")
