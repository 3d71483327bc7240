"""CPU only: which fp16 roundings make up the operand floor?  The fp32 oracle with SELECTED operands rounded to fp16 (linear / conv / GroupNorm inputs,
q k v, P), per hook kind, hook storage rounding included.   python tools/operand_sources.py xl 128   -> profiles/r03_operand_sources_sdxl.txt"""
import sys, os, json, time, contextlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.nn.functional as F
from oracle import unet_ref as R
from oracle.operand_floor import kind_of
torch.set_num_threads(8)
def _r(x): return x.to(torch.float16).to(torch.float32) if x.dtype == torch.float32 and x.dim() >= 2 else x
@contextlib.contextmanager
def rounding(lin_in=False, conv_in=False, gn_in=False, qkv=False, pr=False, vec_rows=64):
    lin, conv, gn, sdpa, mm, sm = F.linear, F.conv2d, F.group_norm, F.scaled_dot_product_attention, torch.matmul, torch.softmax
    def linear(x, w, b=None):
        if x.dim() == 2 and x.shape[0] <= vec_rows: return lin(x, w, b)
        return lin(_r(x) if lin_in else x, w, b)
    def conv2d(x, w, b=None, *a, **k): return conv(_r(x) if conv_in else x, w, b, *a, **k)
    def group_norm(x, *a, **k): return gn(_r(x) if gn_in else x, *a, **k)
    def attention(q, k, v, *a, **kw):
        if qkv: q, k, v = _r(q), _r(k), _r(v)
        scale = kw.get("scale") or q.shape[-1] ** -0.5
        p = sm(mm(q, k.transpose(-1, -2)) * scale, dim=-1)
        return mm(_r(p) if pr else p, v)
    F.linear, F.conv2d, F.group_norm, F.scaled_dot_product_attention = linear, conv2d, group_norm, attention
    try: yield
    finally: F.linear, F.conv2d, F.group_norm, F.scaled_dot_product_attention = lin, conv, gn, sdpa
ver = sys.argv[1]; lat = int(sys.argv[2])
arch = R.ARCHS[ver]
P = R.synth_params(arch, seed=0); I = R.synth_inputs(arch, 1, lat, seed=1)
ids = [i for i in R.stored_hook_ids(arch) if not i.endswith("-map")]
def run():
    st = R.Store({k: True for k in ids}, out_dtype=None)
    with torch.no_grad():
        R.unet_forward(P, arch, I["sample"], I["timestep"], I["ctx"], I.get("text_embeds"), I.get("time_ids"), store=st)
    return st.feats
t0 = time.time(); ref = run(); print("oracle s", time.time() - t0, flush=True)
variants = {
  "all": dict(lin_in=True, conv_in=True, gn_in=True, qkv=True, pr=True),
  "attn_only(qkv+P)": dict(qkv=True, pr=True),
  "P_only": dict(pr=True),
  "qkv_only": dict(qkv=True),
  "gemm_only(lin+conv+gn)": dict(lin_in=True, conv_in=True, gn_in=True),
  "lin_only": dict(lin_in=True),
}
for name, kw in variants.items():
    with rounding(**kw): got = run()
    kinds = {}
    for k in ref:
        # include the fp16 rounding of the hook itself (the product stores fp16 hooks)
        e = float((got[k].half().float() - ref[k]).norm() / ref[k].norm())
        kinds.setdefault(kind_of(k), []).append(e)
    print("==", name, flush=True)
    for kind, v in sorted(kinds.items()):
        print(f"   {kind:16s} n={len(v):3d} median {sorted(v)[len(v)//2]:.2e} worst {max(v):.2e}", flush=True)
