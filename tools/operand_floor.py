#!/usr/bin/env python3
"""fp16-operand floor of the UNet hot path (CPU only, oracle/operand_floor.py): relative L2 error per hook kind of the fp32
oracle run with every matrix-multiply operand rounded to fp16 (everything else fp32) against the plain fp32 oracle.
No fp16-MFMA implementation can be closer to the reference than this; DESIGN.md §4 quotes the table next to the GPU errors.
    python tools/operand_floor.py [--version 1-5|xl] [--lat 64] [--threads 8]"""
import argparse, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from oracle import unet_ref as R
from oracle.operand_floor import fp16_operands, kind_of

ap = argparse.ArgumentParser()
ap.add_argument("--version", default="1-5"); ap.add_argument("--lat", type=int, default=0); ap.add_argument("--threads", type=int, default=0)
ap.add_argument("--tiny", action="store_true")
a = ap.parse_args()
if a.threads: torch.set_num_threads(a.threads)
arch = R.tiny_arch(a.version) if a.tiny else R.ARCHS[a.version]
lat = a.lat or (16 if a.tiny else 128 if a.version == "xl" else 64)
P = R.synth_params(arch, seed=0)
I = R.synth_inputs(arch, 1, lat, seed=1)
ids = [i for i in R.stored_hook_ids(arch) if not i.endswith("-map")]
def run():
    st = R.Store({k: True for k in ids}, out_dtype=None)
    with torch.no_grad():
        R.unet_forward(P, arch, I["sample"], I["timestep"], I["ctx"], I.get("text_embeds"), I.get("time_ids"), store=st)
    return st.feats
t0 = time.time(); ref = run(); t1 = time.time()
with fp16_operands():
    flo = run()
errs = {k: float((flo[k] - ref[k]).norm() / ref[k].norm()) for k in ref}
ev = sorted(errs.values())
print(json.dumps(dict(version=a.version, latent=lat, tiny=a.tiny, oracle_s=round(t1 - t0, 1), hooks=len(ev), p50=ev[len(ev) // 2], worst=ev[-1],
                      below_1e3=sum(e < 1e-3 for e in ev))))
kinds = {}
for k, e in errs.items(): kinds.setdefault(kind_of(k), []).append(e)
for kind, v in sorted(kinds.items()):
    print(f"# floor kind {kind:16s} n={len(v):3d}  median {sorted(v)[len(v) // 2]:.2e}  worst {max(v):.2e}")
