"""Flux arithmetic modes under heavy-tailed weight statistics (companion of tools/heavy_tail_levels.py): FLUX.1-dev WIDTHS (24 heads x 128, 4096 + 512 tokens) with a
reduced stack (4 double + 6 single blocks), batch 1, vs the fp32 oracle — benign N(0, 1/fan_in) weights and heavy-tailed ones (log-normal per-output-channel scales
sigma 0.5 on every linear; 3 outlier channels x16 written into both residual streams by every additive branch: attn.to_out.0 / to_add_out / ff.net.2 /
ff_context.net.2 / single proj_out, and by the embedders).
    python tools/heavy_tail_flux.py"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "generic-diffusion-feature_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import torch  # noqa: E402
from oracle import flux_ref as FR  # noqa: E402
from components.native import NativeFluxTransformer  # noqa: E402

torch.set_num_threads(max(1, min(32, os.cpu_count() or 1)))
arch = dict(FR.ARCH_FLUX_DEV); arch.update(num_layers=4, num_single_layers=6)
C = FR.inner_dim(arch)


def heavy(P, seed=0, sigma=0.5, n_out=3, gain=16.0):
    g = torch.Generator().manual_seed(seed + 7919)
    idx = torch.randperm(C, generator=torch.Generator().manual_seed(1000 + C))[:n_out]
    writers = (".attn.to_out.0.weight", ".attn.to_add_out.weight", ".ff.net.2.weight", ".ff_context.net.2.weight")
    Q = {}
    for name, w in P.items():
        if name.endswith(".weight") and w.dim() == 2:
            sc = torch.exp(sigma * torch.randn(w.shape[0], generator=g))
            if name.endswith(writers) or name in ("x_embedder.weight", "context_embedder.weight") or (name.startswith("single_") and name.endswith(".proj_out.weight")):
                sc[idx] *= gain
            w = w * sc[:, None]
        Q[name] = w.to(torch.bfloat16).float()
    return Q


I = FR.synth_inputs(arch, 1, 64, 512, seed=1)
I = {k: (v.to(torch.bfloat16).float() if k in ("hidden_states", "encoder_hidden_states", "pooled_projections") else v) for k, v in I.items()}
nb = arch["num_layers"] + arch["num_single_layers"]
ids = []
for b in range(nb):
    order = ["q", "k", "v", "attn-out", "norm-out", "ffn-inner", "out"] if b < arch["num_layers"] else ["q", "attn-out", "out"]
    ids += [f"vit-block{b}-{k}" for k in order]
base = {k: v.to(torch.bfloat16).float() for k, v in FR.synth_params(arch, seed=0).items()}
for name, P in (("benign N(0, 1/fan_in)", base), ("heavy-tailed (sigma 0.5, outliers x16)", heavy(base))):
    st = FR.Store({k: True for k in ids})
    t0 = time.time()
    with torch.no_grad():
        y = FR.flux_forward(P, arch, I["hidden_states"], I["encoder_hidden_states"], I["pooled_projections"], I["timestep"], I["img_ids"], I["txt_ids"],
                            I["guidance"], store=st, want_map=False)
    amax = max(float(v.abs().max()) for v in st.feats.values())
    print(f"== Flux widths, {arch['num_layers']} + {arch['num_single_layers']} blocks, {name}: oracle {time.time() - t0:.0f} s; |hook| max {amax:.0f}")
    for dt in ("bfloat16", "auto", "bfloat16x2"):
        net = NativeFluxTransformer(arch, device="cuda:0", compute_dtype=dt)
        net.load_state_dict({k: v.to(torch.bfloat16) for k, v in P.items()})
        out, hooks = net.forward_raw(I["hidden_states"].cuda(), I["encoder_hidden_states"].cuda(), I["pooled_projections"].cuda(), I["timestep"].cuda(),
                                     I["img_ids"].cuda(), I["txt_ids"].cuda(), guidance=I["guidance"].cuda(), hook_ids=ids, grid=(64, 64))
        torch.cuda.synchronize()
        e = {k: float((hooks[k].float().cpu() - st.feats[k].float()).norm() / st.feats[k].float().norm()) for k in ids}
        kinds = {}
        for k, v in e.items():
            kinds.setdefault(k.split("-", 2)[-1], []).append(v)
        worst = max(e, key=e.get)
        print(f"   {dt:11s} (loaded as {net.cfg['compute_dtype']}): worst {worst} = {e[worst]:.2e}; median {sorted(e.values())[len(e) // 2]:.2e}; output "
              f"{float((out.float().cpu() - y).norm() / y.norm()):.2e};  " + "  ".join(f"{kd}={max(v):.1e}" for kd, v in sorted(kinds.items())))
        del net, hooks, out
        torch.cuda.empty_cache()
