#!/usr/bin/env python3
"""verify=True on heavy-tailed weights at TRUE size: which level does the ladder keep, and what does that level measure against the fp32 oracle?
    python tools/verify_heavy_fullsize.py [xl|1-5] [gain ...]"""
import os, sys, time, warnings
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "generic-diffusion-feature_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import torch
from oracle import unet_ref as R
from helpers import cfg_from_oracle_arch
from components.native import NativeUNet

which = sys.argv[1] if len(sys.argv) > 1 else "xl"
gains = [float(x) for x in sys.argv[2:]] or [16.0]
arch = R.ARCHS[which]; lat = 128 if which == "xl" else 64
torch.set_num_threads(max(1, min(32, os.cpu_count() or 1)))
I = R.synth_inputs(arch, 1, lat, seed=1)
ids = [i for i in R.stored_hook_ids(arch) if not i.endswith("-map")]
g = lambda k: I[k].cuda() if k in I else None
for gain in gains:
    P = R.synth_params_heavy(arch, seed=0, outlier_gain=gain)
    with torch.no_grad():
        st = R.Store({k: True for k in ids})
        R.unet_forward(P, arch, I["sample"], I["timestep"], I["ctx"], I.get("text_embeds"), I.get("time_ids"), store=st)
    u = NativeUNet(cfg_from_oracle_arch(arch), device="cuda:0", precise="auto", verify=True)
    u.load_state_dict({k: v.half() for k, v in P.items()})
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        t0 = time.time()
        _, hooks = u.forward_raw(g("sample"), g("timestep"), g("ctx"), g("text_embeds"), g("time_ids"), hook_ids=ids)
        torch.cuda.synchronize()
    errs = {k: float((hooks[k].float().cpu() - st.feats[k].float()).norm() / st.feats[k].float().norm()) for k in ids}
    key, seen, kept = u.verify_log[0]
    print(f"{which} heavy-tailed x{gain:g}: differences to the full split "
          f"{ {m: '%.2e' % v for m, v in seen.items()} } -> kept mask {kept}; kept level vs the fp32 oracle: worst {max(errs.values()):.2e} "
          f"({max(errs, key=errs.get)}), median {sorted(errs.values())[len(errs) // 2]:.2e}; verify took {time.time() - t0:.1f} s", flush=True)
    del u, hooks
    torch.cuda.empty_cache()
