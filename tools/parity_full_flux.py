#!/usr/bin/env python3
"""Full-WIDTH parity check of the MMDiT path: FLUX.1-dev widths (24 heads x 128, 4096 + 512 tokens, T5 width 4096) with a reduced
stack (default 2 double + 3 single blocks so the fp32 CPU oracle finishes in a minute), batch 1, seeded synthetic weights:
native HIP path vs oracle/flux_ref.py (checker only) on every hook.  Prints the relative L2 error per hook.
    python tools/parity_full_flux.py [--layers 2] [--single-layers 3] [--threads 32]"""
import argparse, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "generic-diffusion-feature_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import torch
from oracle import flux_ref as FR
from components.native import NativeFluxTransformer
from helpers import rel_l2

ap = argparse.ArgumentParser()
ap.add_argument("--layers", type=int, default=2); ap.add_argument("--single-layers", type=int, default=3)
ap.add_argument("--threads", type=int, default=32)
a = ap.parse_args()
torch.set_num_threads(min(a.threads, os.cpu_count() or 1))
arch = dict(FR.ARCH_FLUX_DEV); arch.update(num_layers=a.layers, num_single_layers=a.single_layers)
P = FR.synth_params(arch, seed=0)
I = FR.synth_inputs(arch, 1, 64, 512, seed=1)
st = FR.Store(None)
t0 = time.time()
with torch.no_grad():
    y = FR.flux_forward(P, arch, I["hidden_states"], I["encoder_hidden_states"], I["pooled_projections"], I["timestep"],
                        I["img_ids"], I["txt_ids"], I["guidance"], store=st, want_map=False)
t_cpu = time.time() - t0
net = NativeFluxTransformer(arch, device="cuda:0")
net.load_state_dict({k: v.half() for k, v in P.items()})
out, hooks = net.forward_raw(I["hidden_states"].cuda(), I["encoder_hidden_states"].cuda(), I["pooled_projections"].cuda(),
                             I["timestep"].cuda(), I["img_ids"].cuda(), I["txt_ids"].cuda(), guidance=I["guidance"].cuda(),
                             hook_ids=FR.hook_ids(arch), grid=(64, 64))
torch.cuda.synchronize()
errs = {k: rel_l2(hooks[k], st.feats[k]) for k in st.feats}
errs["output"] = rel_l2(out, y)
print(json.dumps(dict(model="flux widths, %d double + %d single blocks, 4096+512 tokens" % (a.layers, a.single_layers),
                      cpu_forward_s=round(t_cpu, 1), worst=max(errs.values()), median=sorted(errs.values())[len(errs) // 2])))
for k, e in errs.items():
    print(f"# {k:28s} {e:.2e}")
# second pass without the pre-norm q / k / v hooks: those blocks take the fused RMSNorm + RoPE epilogue of the QKV GEMM
ids2 = [i for i in FR.hook_ids(arch) if not i.endswith(("-q", "-k", "-v"))]
out2, hooks2 = net.forward_raw(I["hidden_states"].cuda(), I["encoder_hidden_states"].cuda(), I["pooled_projections"].cuda(),
                               I["timestep"].cuda(), I["img_ids"].cuda(), I["txt_ids"].cuda(), guidance=I["guidance"].cuda(),
                               hook_ids=ids2, grid=(64, 64))
torch.cuda.synchronize()
errs2 = {k: rel_l2(hooks2[k], st.feats[k]) for k in ids2}
errs2["output"] = rel_l2(out2, y)
print(json.dumps(dict(pass2="q/k/v un-hooked: RMSNorm + RoPE fused into the QKV GEMM epilogue", worst=max(errs2.values()),
                      median=sorted(errs2.values())[len(errs2) // 2])))
for k, e in errs2.items():
    print(f"# fused {k:22s} {e:.2e}")
