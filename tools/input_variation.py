"""How much does the plain plan's per-hook error move with the INPUTS?  The operand-error table behind the automatic plan chooser was emulated on one seeded
sample / prompt at t = 100.  True widths, batch 1, plain plan vs the fp32 oracle for: the table's own inputs, another sample + prompt at t = 100, the
table's sample at t = 500 and t = 900, another sample at t = 500.  Printed: per case the ratio measured / table over the hooks whose table value is >= 7e-4.
    python tools/input_variation.py [xl|1-5]"""
import os
import sys
import json

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "generic-diffusion-feature_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import torch  # noqa: E402
from oracle import unet_ref as R  # noqa: E402
from helpers import cfg_from_oracle_arch  # noqa: E402
from components.native import NativeUNet, _HERE  # noqa: E402

ver = sys.argv[1] if len(sys.argv) > 1 else "xl"
arch = R.ARCHS[ver]
lat = 128 if ver == "xl" else 64
torch.set_num_threads(max(1, min(32, os.cpu_count() or 1)))
table = json.load(open(os.path.join(_HERE, "operand_error_table.json")))[ver]["hooks"]
P = R.synth_params(arch, seed=0)
ids = [i for i in R.stored_hook_ids(arch) if not i.endswith("-map")]
u = NativeUNet(cfg_from_oracle_arch(arch), device="cuda:0", precise=False)
u.load_state_dict({k: v.half() for k, v in P.items()})
near = [k for k in ids if k in table and table[k][0] >= 7e-4]
out = {}
for name, seed, t in (("table inputs (seed 1, t=100)", 1, 100.0), ("other sample + prompt, t=100", 23, 100.0), ("table sample, t=500", 1, 500.0),
                      ("table sample, t=900", 1, 900.0), ("other sample + prompt, t=500", 23, 500.0), ("third sample + prompt, t=20", 57, 20.0)):
    I = R.synth_inputs(arch, 1, lat, seed=seed)
    I["timestep"] = torch.tensor([t])
    st = R.Store({k: True for k in ids})
    with torch.no_grad():
        R.unet_forward(P, arch, I["sample"], I["timestep"], I["ctx"], I.get("text_embeds"), I.get("time_ids"), store=st)
    g = lambda k: I[k].cuda() if k in I else None
    _, hooks = u.forward_raw(g("sample"), g("timestep"), g("ctx"), g("text_embeds"), g("time_ids"), hook_ids=ids)
    torch.cuda.synchronize()
    errs = {k: float((hooks[k].float().cpu() - st.feats[k].float()).norm() / st.feats[k].float().norm()) for k in ids}
    r = sorted(errs[k] / table[k][0] for k in near)
    worst = max(near, key=lambda k: errs[k] / table[k][0])
    out[name] = errs
    print(f"{ver} {name:32s} measured / table over {len(near)} hooks (table >= 7e-4): median {r[len(r) // 2]:.3f}  p90 {r[int(0.9 * len(r))]:.3f}  max {r[-1]:.3f} ({worst}); "
          f"median error {sorted(errs.values())[len(errs) // 2]:.2e}", flush=True)
    del hooks
if os.environ.get("GDF_DUMP_ERRS"):
    json.dump(out, open(os.path.join(os.environ["GDF_DUMP_ERRS"], f"input_variation_{ver}.json"), "w"))
