"""Companion of tools/input_variation.py for the `*-map` hooks (softmax probabilities: they exponentiate the q / k errors, so their error depends on how peaked the
attention is, i.e. on the inputs): true widths, batch 1, the SELECTIVE plan (what the chooser gives any layer set with a map) vs the fp32 oracle, several inputs.
    python tools/input_variation_maps.py [1-5|xl]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "generic-diffusion-feature_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import torch  # noqa: E402
from oracle import unet_ref as R  # noqa: E402
from helpers import cfg_from_oracle_arch  # noqa: E402
from components.native import NativeUNet, SELECTIVE_BY_ARCH, SPLIT_ALL  # noqa: E402

ver = sys.argv[1] if len(sys.argv) > 1 else "1-5"
arch = R.ARCHS[ver]
lat = 128 if ver == "xl" else 64
torch.set_num_threads(max(1, min(32, os.cpu_count() or 1)))
P = R.synth_params(arch, seed=0)
allids = R.stored_hook_ids(arch)
maps = [i for i in allids if i.endswith("-map")]
if ver == "xl":
    maps = maps[::9]                                   # 16 of the 140 SDXL maps (537 MB each at level 1)
ids = [i for i in allids if i in set(maps)]
for name, seed, t in (("table inputs (seed 1, t=100)", 1, 100.0), ("other sample + prompt, t=100", 23, 100.0), ("table sample, t=500", 1, 500.0),
                      ("table sample, t=900", 1, 900.0), ("third sample + prompt, t=20", 57, 20.0)):
    I = R.synth_inputs(arch, 1, lat, seed=seed)
    I["timestep"] = torch.tensor([t])
    st = R.Store({k: True for k in ids})
    with torch.no_grad():
        R.unet_forward(P, arch, I["sample"], I["timestep"], I["ctx"], I.get("text_embeds"), I.get("time_ids"), store=st, want_map=True)
    g = lambda k: I[k].cuda() if k in I else None
    line = f"{ver} {name:32s} {len(ids)} maps:"
    for lvl, mask in (("selective", SELECTIVE_BY_ARCH[ver]), ("full split", SPLIT_ALL)):
        u = NativeUNet(cfg_from_oracle_arch(arch), device="cuda:0", precise=mask)
        u.load_state_dict({k: v.half() for k, v in P.items()})
        _, hooks = u.forward_raw(g("sample"), g("timestep"), g("ctx"), g("text_embeds"), g("time_ids"), hook_ids=ids)
        torch.cuda.synchronize()
        e = sorted(float((hooks[k].float() - st.feats[k].cuda().float()).norm() / st.feats[k].cuda().float().norm()) for k in ids)
        line += f"  {lvl}: median {e[len(e) // 2]:.2e} worst {e[-1]:.2e}"
        del hooks, u
        torch.cuda.empty_cache()
    print(line, flush=True)
    del st
