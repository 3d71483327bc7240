"""CPU only: the fp16-operand error of the UNet hot path, split by OPERAND CLASS, and the error of any subset of classes kept split (hi + lo).

The fp32 oracle is re-run with exactly one class of matrix-multiply / GroupNorm operands rounded to fp16 at a time (everything else fp32), which gives
each class's variance contribution per hook (the roundings are independent, so variances add: checked against the all-classes run).  A plan that
keeps a SUBSET of classes as split fp16 pairs removes their contributions; `--check a,b,c` re-runs the oracle with everything rounded EXCEPT the
listed classes to confirm the prediction.

    python tools/operand_subsets.py xl 128 --out gpurun_out/subsets_xl.json          # one run per class (~1 min each on 8 cores)
    python tools/operand_subsets.py xl 128 --check attn_out,xq,conv2                 # direct emulation of one selective plan
    python tools/operand_subsets.py 1-5 64 --out gpurun_out/subsets_15.json

Classes (operand = the A side of the contraction; weights are fp16-exact):
  qkv       attn1.to_q/k/v         <- LayerNorm-1 output          attn_out  attn1.to_out.0 / attn2.to_out.0 <- attention output
  xq        attn2.to_q             <- LayerNorm-2 output          geglu     ff.net.0.proj  <- LayerNorm-3 output
  ff_out    ff.net.2               <- GEGLU inner                 proj_in   Transformer2DModel.proj_in  <- GroupNorm output
  proj_out  Transformer2DModel.proj_out <- stream image           conv1 / conv2  ResnetBlock2D convs <- GroupNorm + SiLU output
  shortcut  conv_shortcut (1x1)    <- stream / skip-concat image  sampler   down / upsampler convs <- stream image
  conv_out  conv_out               <- conv_norm_out + SiLU        gn_res1 / gn_res2 / gn_vit / gn_out  fp16 image read by the GroupNorm statistics + apply
  qkv_store q, k, v as stored (flash-attention operands)          P         softmax probabilities as the PV operand
"""
import sys, os, json, time, re, argparse, contextlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.nn.functional as F
from oracle import unet_ref as R
from oracle.operand_floor import kind_of

LIN_CLASSES = [
    ("qkv", r"attn1\.to_[qkv]\.weight$"), ("attn_out", r"attn[12]\.to_out\.0\.weight$"), ("xq", r"attn2\.to_q\.weight$"),
    ("geglu", r"ff\.net\.0\.proj\.weight$"), ("ff_out", r"ff\.net\.2\.weight$"), ("proj_in", r"proj_in\.weight$"),
    ("proj_out", r"proj_out\.weight$"), ("conv1", r"resnets\.\d+\.conv1\.weight$"), ("conv2", r"resnets\.\d+\.conv2\.weight$"),
    ("shortcut", r"conv_shortcut\.weight$"), ("sampler", r"samplers\.0\.conv\.weight$"), ("conv_out", r"^conv_out\.weight$"),
]
# finer classes for follow-up studies (python tools/operand_subsets.py xl 128 --classes attn1_out,attn2_out,upsampler,downsampler); a weight may
# belong to one coarse and one fine class
FINE_CLASSES = [("attn1_out", r"attn1\.to_out\.0\.weight$"), ("attn2_out", r"attn2\.to_out\.0\.weight$"),
                ("upsampler", r"upsamplers\.0\.conv\.weight$"), ("downsampler", r"downsamplers\.0\.conv\.weight$")]
GN_CLASSES = [("gn_res1", r"resnets\.\d+\.norm1\.weight$"), ("gn_res2", r"resnets\.\d+\.norm2\.weight$"),
              ("gn_vit", r"attentions\.\d+\.norm\.weight$"), ("gn_out", r"^conv_norm_out\.weight$")]
ALL = [c for c, _ in LIN_CLASSES] + [c for c, _ in GN_CLASSES] + ["qkv_store", "P"]     # (--classes also takes qkv_self / qkv_cross: the two halves of qkv_store)


def _r(x):
    return x.to(torch.float16).to(torch.float32) if x.dtype == torch.float32 and x.dim() >= 2 else x


def classify(P, fine=False):
    m = {}
    for name, t in P.items():
        for cls, rx in (FINE_CLASSES if fine else LIN_CLASSES + GN_CLASSES):
            if re.search(rx, name):
                m[id(t)] = cls
    return m


@contextlib.contextmanager
def rounding(cls_of, rounded, vec_rows=64):
    lin, conv, gn, sdpa, mm, sm = F.linear, F.conv2d, F.group_norm, F.scaled_dot_product_attention, torch.matmul, torch.softmax

    def linear(x, w, b=None):
        if x.dim() == 2 and x.shape[0] <= vec_rows:
            return lin(x, w, b)
        return lin(_r(x) if cls_of.get(id(w)) in rounded else x, w, b)

    def conv2d(x, w, b=None, *a, **k):
        return conv(_r(x) if cls_of.get(id(w)) in rounded else x, w, b, *a, **k)

    def group_norm(x, g, w=None, b=None, eps=1e-5):
        return gn(_r(x) if cls_of.get(id(w)) in rounded else x, g, w, b, eps)

    def attention(q, k, v, *a, **kw):
        cross = q.shape[-2] != k.shape[-2]             # (the UNets' cross-attention has 77 text keys)
        if "qkv_store" in rounded or ("qkv_cross" if cross else "qkv_self") in rounded:
            q, k, v = _r(q), _r(k), _r(v)
        scale = kw.get("scale") or q.shape[-1] ** -0.5
        p = sm(mm(q, k.transpose(-1, -2)) * scale, dim=-1)
        return mm(_r(p) if "P" in rounded else p, v)

    F.linear, F.conv2d, F.group_norm, F.scaled_dot_product_attention = linear, conv2d, group_norm, attention
    try:
        yield
    finally:
        F.linear, F.conv2d, F.group_norm, F.scaled_dot_product_attention = lin, conv, gn, sdpa


def table(errs, title):
    kinds = {}
    for k, e in errs.items():
        kinds.setdefault(kind_of(k), []).append(e)
    print("==", title, flush=True)
    for kind, v in sorted(kinds.items()):
        print(f"   {kind:16s} n={len(v):3d} median {sorted(v)[len(v)//2]:.2e} worst {max(v):.2e}", flush=True)


# plan operand classes (csrc/builder.h SP_*, components/native.py SPLIT_CLASSES) -> the emulation classes they remove
# (the emulation classes `attn_out` and `sampler` are superseded by their halves attn1_out / attn2_out and downsampler / upsampler, measured in a
# second run with --classes attn1_out,attn2_out,upsampler,downsampler and merged by merge_fine())
PLAN_CLASSES = {"stream": ["shortcut", "proj_out", "gn_vit", "gn_res1", "gn_out"], "gnv": ["proj_in"], "ln_attn": ["qkv", "xq"],
                "attn_out": ["attn1_out"], "attn2_out": ["attn2_out"], "sampler": ["downsampler"], "upsampler": ["upsampler"],
                "ln_ff": ["geglu"], "ff_inner": ["ff_out"], "res": ["conv1", "conv2", "gn_res2"], "out": ["conv_out"],
                "qkv": ["qkv_self"], "xqkv": ["qkv_cross"]}       # (round 5: q / k / v pairs of the self- / cross-attention; the committed r04 runs hold the
                # combined class qkv_store instead: predict() on those keeps it unless BOTH plan classes are split)


def merge_fine(res, fine):
    """replace the coarse classes attn_out / sampler of `res` by the four finer ones of `fine` (same model, same inputs)"""
    out = dict(res); out["classes"] = {c: e for c, e in res["classes"].items() if c not in ("attn_out", "sampler")}
    out["classes"].update(fine["classes"])
    return out


def predict(res, plan_classes):
    """per-hook error (relative L2 vs the fp32 oracle, hook storage included) of a plan that keeps `plan_classes` split: variances add"""
    keep = set(c for p in plan_classes for c in PLAN_CLASSES[p])
    if "qkv_self" in keep and "qkv_cross" in keep:
        keep.add("qkv_store")
    return {h: (res["store"][h] ** 2 + sum(e[h] ** 2 for c, e in res["classes"].items() if c not in keep)) ** 0.5 for h in res["store"]}


def make_table(paths, out):
    """components/operand_error_table.json: per architecture and hook id, the CPU-emulated error of the plain plan and of the selective preset —
    what FeatureExtractor's automatic plan selection consults (components/native.py choose_split)."""
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "generic-diffusion-feature_amd"))
    from components.native import SELECTIVE_BY_ARCH, SPLIT_CLASSES
    tab = {"_meta": {"what": "relative L2 error vs the fp32 oracle per hook: fp32 oracle with the operand classes of the plan rounded to fp16 + fp16 hook "
                             "storage (tools/operand_subsets.py, one run per class, variances added; batch 1, seeded synthetic weights)",
                     "columns": ["plain", "selective"]}}
    for pth in paths:
        res = json.load(open(pth))
        fpth = pth.replace("operand_subsets_", "operand_subsets_fine_").replace("subsets_", "fine_") if "fine" not in pth else None
        for cand in (pth.replace("operand_subsets_", "operand_subsets_fine_"), os.path.join(os.path.dirname(pth), os.path.basename(pth).replace("subsets_", "fine_"))):
            if cand != pth and os.path.exists(cand):
                res = merge_fine(res, json.load(open(cand)))
                break
        else:
            raise SystemExit("no fine-class file next to " + pth)
        ver = res["ver"]
        names = [k for k, b in SPLIT_CLASSES.items() if SELECTIVE_BY_ARCH[ver] & b]
        pl, se = predict(res, []), predict(res, names)
        tab[ver] = {"selective_classes": names, "lat": res["lat"], "hooks": {h: [float("%.3g" % pl[h]), float("%.3g" % se[h])] for h in res["store"]}}
        print(ver, "plain worst %.2e" % max(pl.values()), "selective", names, "worst %.2e" % max(se.values()))
    json.dump(tab, open(out, "w"), separators=(",", ":"))


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "--make-table":
        return make_table(sys.argv[2:-1], sys.argv[-1])
    ap = argparse.ArgumentParser()
    ap.add_argument("ver"); ap.add_argument("lat", type=int)
    ap.add_argument("--out"); ap.add_argument("--check", action="append", default=[])
    ap.add_argument("--threads", type=int, default=8)
    ap.add_argument("--classes", default=",".join(ALL))
    ap.add_argument("--heavy", action="store_true", help="heavy-tailed weights (oracle/unet_ref.py synth_params_heavy, outlier gain 16) instead of N(0, 1/fan_in)")
    a = ap.parse_args()
    torch.set_num_threads(a.threads)
    arch = R.ARCHS[a.ver]
    P = R.synth_params_heavy(arch, seed=0, outlier_gain=16.0) if a.heavy else R.synth_params(arch, seed=0)
    I = R.synth_inputs(arch, 1, a.lat, seed=1)
    ids = [i for i in R.stored_hook_ids(arch) if not i.endswith("-map")]
    fine = any(c in dict(FINE_CLASSES) for c in a.classes.split(","))
    cls_of = classify(P, fine)

    def run(rounded):
        st = R.Store({k: True for k in ids}, out_dtype=None)
        with torch.no_grad(), rounding(cls_of, set(rounded)):
            R.unet_forward(P, arch, I["sample"], I["timestep"], I["ctx"], I.get("text_embeds"), I.get("time_ids"), store=st)
        return st.feats

    def err(got, ref, store16=True):
        return {k: float(((got[k].half().float() if store16 else got[k]) - ref[k]).norm() / ref[k].norm()) for k in ref}

    t0 = time.time(); ref = run(()); print("oracle s %.1f" % (time.time() - t0), flush=True)
    if a.check:
        for spec in a.check:
            keep = [c for c in spec.split(",") if c]
            e = err(run([c for c in ALL if c not in keep]), ref)
            table(e, "split kept for {%s}, everything else fp16" % ",".join(keep))
        return
    res = {"ver": a.ver, "lat": a.lat, "classes": {}, "store": err(ref, ref)}
    table(res["store"], "hook storage only")
    for c in a.classes.split(","):
        t0 = time.time()
        res["classes"][c] = err(run([c]), ref, store16=False)
        table(res["classes"][c], "%s only (no storage rounding)  [%.0f s]" % (c, time.time() - t0))
        if a.out:
            json.dump(res, open(a.out, "w"))
    if not fine:
        res["all"] = err(run(ALL), ref)
        table(res["all"], "all classes + storage")
    if a.out:
        json.dump(res, open(a.out, "w"))


if __name__ == "__main__":
    main()
