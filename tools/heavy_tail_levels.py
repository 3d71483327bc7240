"""Plan levels under heavy-tailed weight statistics: oracle/unet_ref.py synth_params_heavy (log-normal per-channel scales,
a few x`gain` outlier channels in the residual stream) at TRUE widths, batch 1: error of the plain / selective / full-split plans and of the
fp16-operand floor per hook kind, next to the benign N(0, 1/fan_in) weights.
    python tools/heavy_tail_levels.py [xl|1-5|tiny] [lat] [gain]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "generic-diffusion-feature_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import torch  # noqa: E402
from oracle import unet_ref as R  # noqa: E402
from oracle.operand_floor import fp16_operands, kind_of  # noqa: E402
from helpers import cfg_from_oracle_arch  # noqa: E402
from components.native import NativeUNet, SELECTIVE_BY_ARCH, SPLIT_SELECTIVE, arch_family  # noqa: E402

which = sys.argv[1] if len(sys.argv) > 1 else "xl"
arch = R.tiny_arch("xl") if which == "tiny" else R.ARCHS[which]
lat = int(sys.argv[2]) if len(sys.argv) > 2 else (16 if which == "tiny" else 128 if which == "xl" else 64)
gain = float(sys.argv[3]) if len(sys.argv) > 3 else 16.0
torch.set_num_threads(max(1, min(32, os.cpu_count() or 1)))
cfg = cfg_from_oracle_arch(arch)
sel = SELECTIVE_BY_ARCH.get(arch_family(cfg), SPLIT_SELECTIVE)
I = R.synth_inputs(arch, 1, lat, seed=1)
ids = [i for i in R.stored_hook_ids(arch) if not i.endswith("-map")]


def kinds(errs):
    t = {}
    for k, e in errs.items():
        t.setdefault(kind_of(k), []).append(e)
    return "  ".join(f"{kd}={max(v):.2e}" for kd, v in sorted(t.items()))


for name, P in (("benign N(0,1/fan_in)", R.synth_params(arch, 0)), (f"heavy-tailed (sigma 0.5, outliers x{gain:g})", R.synth_params_heavy(arch, 0, outlier_gain=gain))):
    t0 = time.time()
    with torch.no_grad():
        st = R.Store({k: True for k in ids})
        R.unet_forward(P, arch, I["sample"], I["timestep"], I["ctx"], I.get("text_embeds"), I.get("time_ids"), store=st)
        st2 = R.Store({k: True for k in ids})
        with fp16_operands():
            R.unet_forward(P, arch, I["sample"], I["timestep"], I["ctx"], I.get("text_embeds"), I.get("time_ids"), store=st2)
    ref = st.feats
    fl = {k: float((st2.feats[k].float() - ref[k].float()).norm() / ref[k].float().norm()) for k in ids}
    print(f"== {which} lat {lat}, {name}: oracle {time.time() - t0:.0f} s; |hook| max {max(float(v.abs().max()) for v in ref.values()):.0f}")
    print(f"   fp16-operand floor  worst {max(fl.values()):.2e} median {sorted(fl.values())[len(ids) // 2]:.2e}   {kinds(fl)}")
    g = lambda k: I[k].cuda() if k in I else None
    for lvl, spec in (("plain", False), ("selective", sel), ("full split", True)):
        u = NativeUNet(cfg, device="cuda:0", precise=spec)
        u.load_state_dict({k: v.half() for k, v in P.items()})
        _, hooks = u.forward_raw(g("sample"), g("timestep"), g("ctx"), g("text_embeds"), g("time_ids"), hook_ids=ids)
        torch.cuda.synchronize()
        e = {k: float((hooks[k].float().cpu() - ref[k].float()).norm() / ref[k].float().norm()) for k in ids}
        print(f"   {lvl:11s} plan    worst {max(e.values()):.2e} median {sorted(e.values())[len(ids) // 2]:.2e}   {kinds(e)}")
        if lvl == "plain":
            plain_h = {k: hooks[k].float().cpu() for k in ids}
        if lvl == "full split":
            d = {k: float((plain_h[k] - hooks[k].float().cpu()).norm() / hooks[k].float().cpu().norm()) for k in ids}
            print(f"   |plain - full split| (what verify=True measures): worst {max(d.values()):.2e} median {sorted(d.values())[len(ids) // 2]:.2e}")
        del u, hooks
        torch.cuda.empty_cache()
