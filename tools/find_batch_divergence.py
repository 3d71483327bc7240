"""Where do identical samples of one batch stop being bit-identical?  
Runs the true-width UNet on ONE sample repeated B times with every non-map hook and prints, in execution order, the first hooks whose
sample i differs from sample 0: how many elements, and where (rows / channels), which names the op and the mechanism."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "generic-diffusion-feature_amd"))
import torch  # noqa: E402
from components.native import NativeUNet, ARCH_CONFIGS  # noqa: E402

version = sys.argv[1] if len(sys.argv) > 1 else "xl"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 4
lat = int(sys.argv[3]) if len(sys.argv) > 3 else (128 if version == "xl" else 64)
cfg = ARCH_CONFIGS[version]
dev = torch.device("cuda:0")
u = NativeUNet(cfg, device=dev, precise=False).init_synthetic(seed=0)
g = torch.Generator(device=dev).manual_seed(1)
x = torch.randn(1, 4, lat, lat, generator=g, device=dev).half().expand(B, -1, -1, -1).contiguous()
ctx = torch.randn(1, 77, cfg["cross_attention_dim"], generator=g, device=dev).half().expand(B, -1, -1).contiguous()
t = torch.full((B,), 100.0, device=dev)
txt = tid = None
if cfg["addition_embed_text_time"]:
    pooled = cfg["add_in_dim"] - 6 * cfg["addition_time_embed_dim"]
    txt = torch.randn(1, pooled, generator=g, device=dev).half().expand(B, -1).contiguous()
    tid = torch.tensor([[lat * 8, lat * 8, 0, 0, lat * 8, lat * 8]], dtype=torch.float32, device=dev).repeat(B, 1)
ids = [i for i in u.hook_names() if not i.endswith("-map")]
for shared in (True, False):
    _, hooks = u.forward_raw(x, t, ctx, txt, tid, hook_ids=ids, shared_ctx=shared)
    torch.cuda.synchronize()
    shown = 0
    total = 0
    for k in ids:
        h = hooks[k]
        nd = [int((h[i] != h[0]).sum()) for i in range(1, B)]
        if any(nd):
            total += 1
            if shown < 6:
                shown += 1
                i = 1 + max(range(B - 1), key=lambda j: nd[j])
                d = (h[i] != h[0])
                ch = d.flatten(1).any(1).nonzero().flatten()
                rows = d.any(0).flatten().nonzero().flatten()
                print(f"[shared_ctx={shared}] {k} shape {tuple(h.shape)}: differing elements per sample {nd}; sample {i}: {int(d.sum())} elements in "
                      f"{len(ch)} channels (first {ch[:6].tolist()}), {len(rows)} pixels (first {rows[:6].tolist()}); max |diff| "
                      f"{float((h[i].float() - h[0].float()).abs().max()):.3e}")
    print(f"[shared_ctx={shared}] hooks with any difference: {total} of {len(ids)}")
    del hooks
