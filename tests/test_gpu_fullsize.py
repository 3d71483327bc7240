"""FULL-SIZE parity (-m gpu): the TRUE SDXL / SD1.5 / Flux.1-dev widths at the BASELINE.json batch sizes, through the C ABI,
against the fp32 CPU oracle on identical seeded weights / latents / timestep / prompt-embeds.

Reference contract: every `feature_gatherer.gather` site of unet_2d_condition.py:1040-1319 (+ resnet.py, attention.py,
attention_processor.py, transformer_2d.py) and of transformer_flux.py:414-603.  Metric: relative L2 error per hooked tensor.

Stated tolerances (north star: 1e-3):
  * SDXL 1024^2, every hook kind                  <= 1.0e-3, except `ffn-inner` and `unet-out`      <= 1.3e-3
  * SD1.5 512^2 (narrower layers average less)    <= 1.1e-3, except `ffn-inner` and `unet-out`      <= 1.35e-3;
    `*-map` (softmax of q k^T: exponentiates the q / k errors)                                        <= 1.5e-3
  * Flux widths: compute_dtype float16            <= 6e-4;   bfloat16 (the reference's dtype, 8 mantissa bits) <= 4e-3
  * SDXL config_xl_full maps (140 `*-map` ids, B = 1) <= 1.5e-3;  PixArt-Sigma widths <= 6e-4;  VAE encoder 1024^2 latents <= 1e-3
  * PRECISE plans (opt-in, NativeUNet(precise=True) / GDF_PRECISE=1: split fp16 hi + lo activation operands, K doubled; round 5: q / k / v of
    both attentions as pairs through the flash kernel): EVERY hook kind incl. `ffn-inner`, `unet-out`      <= 3e-4 (SDXL) / 6e-4 (SD1.5 incl. maps)
  * and for the UNets: every hook within 1.15x (+2e-5) of the fp16-OPERAND FLOOR (oracle/operand_floor.py) — the error of
    the fp32 oracle with nothing but its matmul operands rounded to fp16, i.e. what any fp16-MFMA implementation commits at
    best.  `ffn-inner` (h * gelu(g): the product of two GEMM outputs that each carry the stream error) and `unet-out` sit
    above 1e-3 ON THAT FLOOR (profiles/r02_operand_floor_*.txt), which is why they have their own bound.
The oracle runs at batch 1 (2 for SD1.5); the GPU runs at the BASELINE batch (SDXL 16, SD1.5 32, Flux 8) on the same
sample repeated, so the tile selection of configs C2 / C3 / C5 is the one that is checked, and EVERY sample is compared.
"""
import os

import pytest
import torch

from helpers import cfg_from_oracle_arch
from oracle import unet_ref as R
from oracle.operand_floor import fp16_operands, kind_of

pytestmark = pytest.mark.gpu


def _threads():
    torch.set_num_threads(max(1, min(32, os.cpu_count() or 1)))


def _oracle(arch, P, I, ids, floor=False, want_map=None):
    st = R.Store({k: True for k in ids})
    with torch.no_grad():
        if floor:
            with fp16_operands():
                R.unet_forward(P, arch, I["sample"], I["timestep"], I["ctx"], I.get("text_embeds"), I.get("time_ids"), store=st,
                               want_map=want_map)
        else:
            R.unet_forward(P, arch, I["sample"], I["timestep"], I["ctx"], I.get("text_embeds"), I.get("time_ids"), store=st,
                           want_map=want_map)
    return st.feats


def _rel_each(hook, ref):
    """relative L2 error of every sample of `hook` (Bg, ...) against ref (Br, ...) (sample i vs ref[i % Br]), on the GPU"""
    r = ref.cuda().float()
    rn = r.flatten(1).norm(dim=1)
    out = []
    for i in range(hook.shape[0]):
        j = i % r.shape[0]
        out.append(float((hook[i].float() - r[j]).norm() / rn[j]))
    return out


def _native(arch, P, precise=False):
    from components.native import NativeUNet
    u = NativeUNet(cfg_from_oracle_arch(arch), device="cuda:0", precise=precise)
    u.load_state_dict({k: v.half() for k, v in P.items()})
    return u


def _rep(I, B):
    out = {}
    for k, v in I.items():
        out[k] = v if k == "timestep" else v[:1].expand(B, *v.shape[1:]).contiguous()
    return out


def _margin(tag, errs, bound, extra=""):
    """record the hook with the least spare room (err / bound) of a configuration for the driver-visible summary (tests/conftest.py)"""
    import conftest
    b = bound if callable(bound) else (lambda kd, _b=bound: _b)
    k = max(errs, key=lambda h: errs[h] / b(kind_of(h)))
    conftest.record_margin(tag, k, errs[k], b(kind_of(k)), extra)


def _check(errs, floor, bound, floor_mult=1.15, floor_abs=2e-5, tag=None):
    """errs / floor: {id: worst-over-samples error}; bound(kind) -> absolute tolerance"""
    kinds = {}
    bad = []
    if tag:
        _margin(tag, errs, bound)
    for k, e in errs.items():
        kd = kind_of(k)
        kinds.setdefault(kd, []).append(e)
        if not e <= bound(kd):
            bad.append((k, e, "bound", bound(kd)))
        if floor is not None and not e <= floor_mult * floor[k] + floor_abs:
            bad.append((k, e, "floor", floor[k]))
    for kd, v in sorted(kinds.items()):
        fl = [floor[k] for k in errs if kind_of(k) == kd] if floor is not None else [0.0]
        print(f"  kind {kd:16s} n={len(v):3d}  gpu median {sorted(v)[len(v) // 2]:.2e} worst {max(v):.2e}   floor worst {max(fl):.2e}")
    assert not bad, bad[:10]


def _plain_plan_contract(fam, arch, errs, tag, level_mask=0, product=True):
    """the automatic chooser hands a hook to the plan level `level_mask` (0 = the PLAIN plan, SPLIT_LIGHT = only the `gnv`
    class split) when its emulated error under that level (components/operand_error_table.json) is <= AUTO_BOUND.  `errs` = measured error of
    every hook under that level on hardware.  Asserted: EVERY hook h with choose_split([h]) == level_mask — i.e. every hook a user may request
    alone and get this level for — is within 1e-3 with >= 3 % to spare; printed: the measured hardware offset over the emulation for the
    hooks near the bound (max errs / table), the number AUTO_BOUND is derived from."""
    import json
    from components.native import choose_split, table_scale, _HERE
    import components.native as _N
    cfg = cfg_from_oracle_arch(arch)
    AUTO_BOUND = _N.auto_bound(cfg) / table_scale(cfg)          # the bound this MODEL's table values are compared with (SD2.1 borrows SD1.5's table x 1.04)
    table = json.load(open(os.path.join(_HERE, "operand_error_table.json")))[fam]["hooks"]
    col = 0 if level_mask == 0 else 2
    accepted = [h for h in errs if not h.endswith("-map") and choose_split(cfg, [h]) == level_mask]
    assert len(accepted) > (len(errs) // 4 if level_mask == 0 else 0)
    ratio = {h: errs[h] / table[h][col] for h in accepted if h in table and table[h][col] >= 8e-4}      # the hooks NEAR the bound decide
    hmax = max(ratio, key=ratio.get)
    worst = max(accepted, key=lambda h: errs[h])
    print(f"[{tag}] plan level {level_mask}: {len(accepted)} of {len(errs)} hooks are handed to this level at AUTO_BOUND {AUTO_BOUND:.2e}; "
          f"hardware / emulation offset near the bound: max {ratio[hmax]:.3f} ({hmax}), median {sorted(ratio.values())[len(ratio) // 2]:.3f}; "
          f"worst accepted hook {worst} = {errs[worst]:.2e}")
    dump = os.environ.get("GDF_DUMP_ERRS")
    if dump:
        json.dump({"errs": errs, "accepted": accepted, "auto_bound": AUTO_BOUND}, open(os.path.join(dump, f"plan_level_{level_mask}_errs_{tag}.json"), "w"))
    import conftest
    conftest.record_margin(f"plan-level contract [{tag}] level {level_mask}: worst hook the chooser hands to this level ALONE", worst, errs[worst], 0.97e-3,
                           extra=f"{len(accepted)} hooks at AUTO_BOUND {AUTO_BOUND:.2e}; hardware / emulation offset max {ratio[hmax]:.3f}")
    bad = [(h, errs[h]) for h in accepted if not errs[h] <= 0.97e-3]
    assert not bad, sorted(bad, key=lambda kv: -kv[1])[:10]
    # on the table's own kind of inputs: the bound leaves >= 3 % to 1e-3 even if the worst offset met the largest accepted table value
    # (product=False: inputs chosen to differ from the table's, where the per-hook assertion above is the contract and the offsets are what is being measured)
    assert not product or AUTO_BOUND * max(ratio.values()) <= 0.97e-3, (AUTO_BOUND, max(ratio.values()))


def _light_level_errs(arch, P, run, ref, ids):
    """the same batch under the LIGHT level (SPLIT_LIGHT: only the GroupNorm output in front of proj_in split) -> {hook: worst sample error}"""
    from components.native import SPLIT_LIGHT
    ul = _native(arch, P, precise=SPLIT_LIGHT)
    hooks = run(ul)
    torch.cuda.synchronize()
    assert ul.last_split == SPLIT_LIGHT
    out = {k: max(_rel_each(hooks[k], ref[k])) for k in ids}
    del hooks, ul
    torch.cuda.empty_cache()
    return out


def test_sdxl_1024_batch16_all_non_map_hooks():
    """BASELINE config C3 (SDXL 1024^2, B = 16): all 472 non-map ids of config_xl_full, every sample."""
    _threads()
    arch = R.ARCHS["xl"]
    P = R.synth_params(arch, seed=0)
    I = R.synth_inputs(arch, 1, 128, seed=1)
    ids = [i for i in R.stored_hook_ids(arch) if not i.endswith("-map")]
    assert len(ids) == 472
    ref = _oracle(arch, P, I, ids)
    floor_f = _oracle(arch, P, I, ids, floor=True)
    floor = {k: float((floor_f[k].float() - ref[k].float()).norm() / ref[k].float().norm()) for k in ids}
    del floor_f
    u = _native(arch, P)
    B = 16
    Ib = _rep(I, B)
    g = lambda k: Ib[k].cuda()
    noise, hooks = u.forward_raw(g("sample"), g("timestep"), g("ctx"), g("text_embeds"), g("time_ids"), hook_ids=ids, shared_ctx=True)
    torch.cuda.synchronize()
    assert list(hooks.keys()) == ids
    errs, n_differ, first_differ = {}, 0, None
    for k in ids:
        assert tuple(hooks[k].shape[1:]) == tuple(ref[k].shape[1:]) and hooks[k].shape[0] == B and hooks[k].dtype == torch.float16
        e = _rel_each(hooks[k], ref[k])
        errs[k] = max(e)
        nd = sum(1 for i in range(1, B) if not torch.equal(hooks[k][i], hooks[k][0]))
        if nd and first_differ is None:
            first_differ = k
        n_differ += nd
    ev = sorted(errs.values())
    print(f"\n[sdxl 1024^2 B=16] hooks={len(ev)} median {ev[len(ev) // 2]:.2e} worst {ev[-1]:.2e}; below 1e-3: {sum(e < 1e-3 for e in ev)}; "
          f"(sample, hook) pairs not bit-identical to sample 0: {n_differ} (first: {first_differ})")
    _check(errs, floor, lambda kd: 1.3e-3 if kd in ("ffn-inner", "unet-out") else 1.0e-3, tag="SDXL 1024^2 B=16 plain plan, 472 non-map hooks (ffn-inner / unet-out: 1.3e-3)")
    _plain_plan_contract("xl", arch, errs, "sdxl_b16")
    assert n_differ == 0, (n_differ, first_differ)       # identical samples give identical bits at every batch position (round 5: small_linear_wide fix)
    from components.native import SPLIT_LIGHT
    errs_l = _light_level_errs(arch, P, lambda ul: ul.forward_raw(g("sample"), g("timestep"), g("ctx"), g("text_embeds"), g("time_ids"),
                                                                  hook_ids=ids, shared_ctx=True)[1], ref, ids)
    _plain_plan_contract("xl", arch, errs_l, "sdxl_b16", SPLIT_LIGHT)
    # the four practical hooks of the headline bench (config_xl_practical) meet the north star with margin
    for k in ("up-level0-repeat0-vit-block7-out", "up-level0-repeat0-vit-block5-out", "up-level1-repeat0-vit-block0-cross-q",
              "up-level1-repeat0-vit-block0-out"):
        assert errs[k] < 9.5e-4, (k, errs[k])
    # ---- the same batch through a PRECISE plan (split hi + lo operands): the north-star bound on EVERY kind, measured on hardware ----
    del hooks, noise, u
    torch.cuda.empty_cache()
    up = _native(arch, P, precise=True)
    _, hooks = up.forward_raw(g("sample"), g("timestep"), g("ctx"), g("text_embeds"), g("time_ids"), hook_ids=ids, shared_ctx=True)
    torch.cuda.synchronize()
    errs_p = {k: max(_rel_each(hooks[k], ref[k])) for k in ids}
    ev = sorted(errs_p.values())
    print(f"[sdxl 1024^2 B=16 PRECISE] hooks={len(ev)} median {ev[len(ev) // 2]:.2e} worst {ev[-1]:.2e}; below 1e-3: {sum(e < 1e-3 for e in ev)}")
    _check(errs_p, None, lambda kd: 3.0e-4, tag="SDXL 1024^2 B=16 PRECISE (full split)")        # (round 5: q / k / v pairs in the full split: worst 1.9e-4; 4.9e-4 before)
    # ---- the PRODUCT DEFAULT ('auto', round 4): the plan level is chosen from the requested hooks — this set contains `ffn-inner` /
    # `unet-out`, so the selective split (stream images + GroupNorm-in-front-of-proj_in + attention outputs + conv_out operand) is picked,
    # and EVERY kind meets the north-star 1e-3 at ~0.9x the plain plan's speed (tools/bench_split.py); the headline's four hooks alone
    # select the plain plan
    del hooks, up
    torch.cuda.empty_cache()
    from components.native import SELECTIVE_BY_ARCH
    ua = _native(arch, P, precise="auto")
    assert ua.split_for(["up-level0-repeat0-vit-block7-out", "up-level0-repeat0-vit-block5-out", "up-level1-repeat0-vit-block0-cross-q",
                         "up-level1-repeat0-vit-block0-out"]) == 0
    _, hooks = ua.forward_raw(g("sample"), g("timestep"), g("ctx"), g("text_embeds"), g("time_ids"), hook_ids=ids, shared_ctx=True)
    torch.cuda.synchronize()
    assert ua.last_split == SELECTIVE_BY_ARCH["xl"]
    errs_a = {k: max(_rel_each(hooks[k], ref[k])) for k in ids}
    ev = sorted(errs_a.values())
    print(f"[sdxl 1024^2 B=16 AUTO -> selective split] hooks={len(ev)} median {ev[len(ev) // 2]:.2e} worst {ev[-1]:.2e}; below 1e-3: {sum(e < 1e-3 for e in ev)}")
    _check(errs_a, None, lambda kd: 1.0e-3, tag="SDXL 1024^2 B=16 AUTO -> selective split, 472 hooks")


def test_sd15_512_full_layer_set_with_maps_batch2_and_batch32():
    """BASELINE config C2 (SD1.5 512^2, full 197-id layer set incl. attention maps): B = 2 on two different samples and
    prompts against the oracle, then the B = 32 plan on sample 0 repeated (every sample compared)."""
    _threads()
    arch = R.ARCHS["1-5"]
    P = R.synth_params(arch, seed=0)
    I = R.synth_inputs(arch, 2, 64, seed=1, same_prompt=False)
    ids = R.stored_hook_ids(arch)
    assert len(ids) == 197
    ref = _oracle(arch, P, I, ids, want_map=True)
    nm = [i for i in ids if not i.endswith("-map")]
    floor_f = _oracle(arch, P, I, ids, floor=True, want_map=True)
    floor = {k: float((floor_f[k].float() - ref[k].float()).norm() / ref[k].float().norm()) for k in ids}
    del floor_f
    u = _native(arch, P)
    g = lambda k: I[k].cuda()
    _, hooks = u.forward_raw(g("sample"), g("timestep"), g("ctx"), hook_ids=ids)
    torch.cuda.synchronize()
    assert list(hooks.keys()) == ids
    bound = lambda kd: 1.35e-3 if kd in ("ffn-inner", "unet-out") else (1.5e-3 if kd == "map" else 1.1e-3)
    errs = {k: max(_rel_each(hooks[k], ref[k])) for k in ids}
    ev = sorted(errs.values())
    print(f"\n[sd1.5 512^2 B=2, 197 ids] median {ev[len(ev) // 2]:.2e} worst {ev[-1]:.2e}; below 1e-3: {sum(e < 1e-3 for e in ev)}")
    _check(errs, floor, bound, tag="SD1.5 512^2 B=2 plain plan, 197 ids incl. maps")
    del hooks
    torch.cuda.empty_cache()
    # ---- the batch-32 plan of config C2 (56 GB of hooks) on sample 0 repeated ----
    B = 32
    I0 = {k: (v if k == "timestep" else v[:1]) for k, v in I.items()}
    Ib = _rep(I0, B)
    gb = lambda k: Ib[k].cuda()
    _, hooks = u.forward_raw(gb("sample"), gb("timestep"), gb("ctx"), hook_ids=ids, shared_ctx=True)
    torch.cuda.synchronize()
    errs32 = {}
    for k in ids:
        assert hooks[k].shape[0] == B
        errs32[k] = max(_rel_each(hooks[k], ref[k][:1]))
    ev = sorted(errs32.values())
    print(f"[sd1.5 512^2 B=32, 197 ids] median {ev[len(ev) // 2]:.2e} worst {ev[-1]:.2e}")
    floor0 = None      # (the floor above was taken over both samples; the absolute bounds are what is asserted at B = 32)
    _check({k: errs32[k] for k in nm}, floor0, bound, tag="SD1.5 512^2 B=32 plain plan, non-map hooks")
    _check({k: errs32[k] for k in ids if k.endswith("-map")}, None, bound, tag="SD1.5 512^2 B=32 plain plan, maps")
    _plain_plan_contract("1-5", arch, {k: errs[k] for k in nm}, "sd15_b2")
    _plain_plan_contract("1-5", arch, {k: errs32[k] for k in nm}, "sd15_b32")
    from components.native import SPLIT_LIGHT
    del hooks
    torch.cuda.empty_cache()
    errs_l = _light_level_errs(arch, P, lambda ul: ul.forward_raw(gb("sample"), gb("timestep"), gb("ctx"), hook_ids=nm, shared_ctx=True)[1],
                               {k: ref[k][:1] for k in nm}, nm)
    _plain_plan_contract("1-5", arch, errs_l, "sd15_b32", SPLIT_LIGHT)
    hooks = None
    # ---- PRECISE plans (split hi + lo operands): every kind incl. `ffn-inner`, `unet-out` and the maps <= 1e-3, B = 2 and B = 32 ----
    del hooks, u
    torch.cuda.empty_cache()
    up = _native(arch, P, precise=True)
    _, hooks = up.forward_raw(g("sample"), g("timestep"), g("ctx"), hook_ids=ids)
    torch.cuda.synchronize()
    errs_p = {k: max(_rel_each(hooks[k], ref[k])) for k in ids}
    ev = sorted(errs_p.values())
    print(f"[sd1.5 512^2 B=2 PRECISE, 197 ids] median {ev[len(ev) // 2]:.2e} worst {ev[-1]:.2e}; below 1e-3: {sum(e < 1e-3 for e in ev)}")
    _check(errs_p, None, lambda kd: 6.0e-4, tag="SD1.5 512^2 B=2 PRECISE, 197 ids")        # (incl. the maps, whose kernel reads the hi halves of q / k: worst 4.4e-4)
    del hooks
    torch.cuda.empty_cache()
    _, hooks = up.forward_raw(gb("sample"), gb("timestep"), gb("ctx"), hook_ids=ids, shared_ctx=True)
    torch.cuda.synchronize()
    errs_p32 = {k: max(_rel_each(hooks[k], ref[k][:1])) for k in ids}
    ev = sorted(errs_p32.values())
    print(f"[sd1.5 512^2 B=32 PRECISE, 197 ids] median {ev[len(ev) // 2]:.2e} worst {ev[-1]:.2e}")
    _check(errs_p32, None, lambda kd: 6.0e-4, tag="SD1.5 512^2 B=32 PRECISE, 197 ids")
    # ---- the PRODUCT DEFAULT ('auto'): this full layer set (maps, ffn-inner, unet-out) selects the SD1.5 selective split; every kind <= 1e-3 ----
    del hooks, up
    torch.cuda.empty_cache()
    from components.native import SELECTIVE_BY_ARCH
    ua = _native(arch, P, precise="auto")
    _, hooks = ua.forward_raw(g("sample"), g("timestep"), g("ctx"), hook_ids=ids)
    torch.cuda.synchronize()
    assert ua.last_split == SELECTIVE_BY_ARCH["1-5"]
    errs_a = {k: max(_rel_each(hooks[k], ref[k])) for k in ids}
    ev = sorted(errs_a.values())
    print(f"[sd1.5 512^2 B=2 AUTO -> selective split, 197 ids] median {ev[len(ev) // 2]:.2e} worst {ev[-1]:.2e}; below 1e-3: {sum(e < 1e-3 for e in ev)}")
    _check(errs_a, None, lambda kd: 1.0e-3, tag="SD1.5 512^2 B=2 AUTO -> selective split, 197 ids")
    del hooks
    torch.cuda.empty_cache()
    _, hooks = ua.forward_raw(gb("sample"), gb("timestep"), gb("ctx"), hook_ids=ids, shared_ctx=True)
    torch.cuda.synchronize()
    errs_a32 = {k: max(_rel_each(hooks[k], ref[k][:1])) for k in ids}
    ev = sorted(errs_a32.values())
    print(f"[sd1.5 512^2 B=32 AUTO -> selective split, 197 ids] median {ev[len(ev) // 2]:.2e} worst {ev[-1]:.2e}")
    _check(errs_a32, None, lambda kd: 1.0e-3, tag="SD1.5 512^2 B=32 AUTO -> selective split, 197 ids")


@pytest.mark.parametrize("dt", ["bfloat16", "float16", "bfloat16x2", "float16s"])
def test_flux_full_width_batch8(dt):
    """BASELINE config C5 widths (24 heads x 128, 4096 + 512 tokens, T5 width 4096) with a reduced stack (2 double + 3 single
    blocks so the CPU oracle finishes within a minute), batch 8 on one sample repeated: every non-map hook, both QKV paths
    (pre-norm q / k / v hooked: separate RMSNorm + RoPE pass; un-hooked: fused into the QKV GEMM epilogue)."""
    _threads()
    from oracle import flux_ref as FR
    from components.native import NativeFluxTransformer
    arch = dict(FR.ARCH_FLUX_DEV); arch.update(num_layers=2, num_single_layers=3)
    P = FR.synth_params(arch, seed=0)
    I = FR.synth_inputs(arch, 1, 64, 512, seed=1)
    tdt = torch.float16 if dt == "float16" else torch.bfloat16
    P = {k: v.to(tdt).float() for k, v in P.items()}            # the oracle runs on the values the model's element type holds
    I = {k: (v.to(tdt).float() if k in ("hidden_states", "encoder_hidden_states", "pooled_projections") else v) for k, v in I.items()}
    tol = {"bfloat16": 4e-3, "float16": 6e-4, "bfloat16x2": 1e-3, "float16s": 6e-4}[dt]
    st = FR.Store(None)
    with torch.no_grad():
        y = FR.flux_forward(P, arch, I["hidden_states"], I["encoder_hidden_states"], I["pooled_projections"], I["timestep"],
                            I["img_ids"], I["txt_ids"], I["guidance"], store=st, want_map=False)
    net = NativeFluxTransformer(arch, device="cuda:0", compute_dtype=dt)
    net.load_state_dict({k: v.to(tdt) for k, v in P.items()})
    B = 8
    rep = lambda t: t[:1].expand(B, *t.shape[1:]).contiguous().cuda()
    args = (rep(I["hidden_states"]), rep(I["encoder_hidden_states"]), rep(I["pooled_projections"]), I["timestep"].cuda(),
            I["img_ids"].cuda(), I["txt_ids"].cuda())
    all_ids = FR.hook_ids(arch)
    for ids in (all_ids, [i for i in all_ids if not i.endswith(("-q", "-k", "-v"))]):
        out, hooks = net.forward_raw(*args, guidance=I["guidance"].cuda(), hook_ids=ids, grid=(64, 64))
        torch.cuda.synchronize()
        assert list(hooks.keys()) == ids
        errs = {k: max(_rel_each(hooks[k], st.feats[k])) for k in ids}
        e_out = max(_rel_each(out, y))
        if dt == "bfloat16x2":                                  # the model OUTPUT is a bf16 tensor (8 mantissa bits of storage): its own bound
            assert e_out <= 3e-3, e_out
        else:
            errs["output"] = e_out
        worst = max(errs, key=errs.get)
        print(f"\n[flux widths B=8 {dt}, {len(ids)} hooks] worst {worst} = {errs[worst]:.2e}; output {e_out:.2e}")
        if len(ids) == len(all_ids):
            _margin(f"Flux widths 2+3 blocks B=8 {dt}, {len(ids)} hooks", errs, tol)
        assert errs[worst] <= tol, (worst, errs[worst])
        del hooks, out


def test_flux_dev_full_depth_config_c5_error_vs_depth():
    """BASELINE config C5 at FULL DEPTH: FLUX.1-dev's 19 double + 38 single blocks, 24 heads x 128, 4096 image + 512 text tokens,
    batch 8 on the GPU (one sample repeated) against ONE batch-1 run of the fp32 CPU oracle (74.4 TFLOP), in the reference's bf16
    and in fp16.  Reference contract: FluxTransformer2DModel.forward, transformer_flux.py:414-603.  Hooks: every block's `out`
    (error vs depth; the double blocks' `out` is the modulated LayerNorm output, :200-211) plus q / attn-out / ffn-inner of a few
    blocks and the model output.  Weights are generated per tensor on demand (helpers.LazySynthParams) with values exact in both
    element types.  The per-depth table is printed (and written to gpurun_out/flux_depth_parity.txt when that directory exists)."""
    import time
    from oracle import flux_ref as FR
    from components.native import NativeFluxTransformer
    from helpers import LazySynthParams, _DeviceView
    _threads()                                     # 32 threads: more OpenMP threads are SLOWER on the 256-CPU hosts (bench.py cpu_baseline)
    arch = dict(FR.ARCH_FLUX_DEV)
    nd, ns = arch["num_layers"], arch["num_single_layers"]
    P = LazySynthParams(FR.param_shapes(arch), lambda n: ".attn.norm_" in n, device="cuda:0", seed=0)
    I = FR.synth_inputs(arch, 1, 64, 512, seed=1)
    I = {k: (v.to(torch.bfloat16).float() if k in ("hidden_states", "encoder_hidden_states", "pooled_projections") else v) for k, v in I.items()}
    ids = []
    for b in range(nd + ns):
        extra = ["q", "attn-out"] + (["ffn-inner"] if b < nd else []) if b in (0, nd - 1, nd, nd + ns // 2, nd + ns - 1) else []
        order = ["q", "attn-out", "norm-out", "ffn-inner", "out"] if b < nd else ["q", "attn-out", "out"]
        ids += [f"vit-block{b}-{k}" for k in order if k == "out" or k in extra]
    st = FR.Store({k: True for k in ids})
    t0 = time.time()
    with torch.no_grad():
        y = FR.flux_forward(P, arch, I["hidden_states"], I["encoder_hidden_states"], I["pooled_projections"], I["timestep"],
                            I["img_ids"], I["txt_ids"], I["guidance"], store=st, want_map=False)
    t_oracle = time.time() - t0
    assert list(st.feats.keys()) == ids
    B = 8
    rep = lambda t: t[:1].expand(B, *t.shape[1:]).contiguous().cuda()
    lines = [f"FLUX.1-dev full depth ({nd} double + {ns} single blocks), 4096 + 512 tokens, GPU batch {B} vs fp32 oracle batch 1 "
             f"(oracle {t_oracle:.0f} s on {torch.get_num_threads()} threads); relative L2 error, worst sample"]
    # measured (profiles/r03_flux_depth_parity.txt): the error saturates with depth — block `out` 1.7e-3 (block 0) -> 2.65e-3 (block 56) in
    # bf16, 1.3e-4 -> 4.3e-4 in fp16; worst hook (q of the last block) 3.4e-3 / 4.6e-4
    # round 4: 'bfloat16x2' (bf16 hi + lo operand pairs, fp16 attention internals): every HOOK within the north-star 1e-3 at full depth
    # without leaving bf16's range on the residual / MLP path (the model output is a bf16 tensor: storage-limited, its own bound)
    # 'fp8-mx' (opt-in, LOWER precision than the reference's bf16; BASELINE.json configs[4] "optional fp8 MFMA"): the large linears on e4m3
    # operands; its own stated bound, error-vs-depth table below
    # round 5, 'auto' (the product default = 'float16s' behind the load-time weight-cast guard): fp16 operands, MLP hidden tensors range-scaled by 2^-8,
    # a bf16 CHECKPOINT handed in: every hook AND the output within the north-star 1e-3 at the bf16 mode's speed
    bounds = {"bfloat16": 4.0e-3, "float16": 6.0e-4, "bfloat16x2": 1.0e-3, "fp8-mx": 1.0e-1, "auto": 8.5e-4}
    for dt, tdt in (("bfloat16", torch.bfloat16), ("float16", torch.float16), ("bfloat16x2", torch.bfloat16), ("fp8-mx", torch.bfloat16),
                    ("auto", torch.bfloat16)):
        net = NativeFluxTransformer(arch, device="cuda:0", compute_dtype=dt)
        net.load_state_dict(_DeviceView(P, tdt))
        args = (rep(I["hidden_states"]), rep(I["encoder_hidden_states"]), rep(I["pooled_projections"]), I["timestep"].cuda(),
                I["img_ids"].cuda(), I["txt_ids"].cuda())
        out, hooks = net.forward_raw(*args, guidance=I["guidance"].cuda(), hook_ids=ids, grid=(64, 64))
        torch.cuda.synchronize()
        assert list(hooks.keys()) == ids
        errs = {k: max(_rel_each(hooks[k], st.feats[k])) for k in ids}
        errs["output"] = max(_rel_each(out, y))
        e_out = errs["output"]
        if dt == "bfloat16x2":
            assert errs.pop("output") <= 3e-3
            errs["output"] = 0.0
        for k in ids:
            assert torch.isfinite(hooks[k].float()).all(), k
        outs = [errs[f"vit-block{b}-out"] for b in range(nd + ns)]
        lines.append(f"[{dt}] block `out` error by depth: " + " ".join(f"{b}:{e:.2e}" for b, e in enumerate(outs)))
        lines.append(f"[{dt}] other hooks: " + " ".join(f"{k}:{e:.2e}" for k, e in errs.items() if not k.endswith("-out") or k == "output" or "attn" in k))
        worst = max(errs, key=errs.get)
        lines.append(f"[{dt}] worst {worst} = {errs[worst]:.2e} (bound {bounds[dt]:.1e}); model output {e_out:.2e}")
        print("\n" + "\n".join(lines[-3:]))
        _margin(f"FLUX.1-dev FULL depth 19+38, 4096+512 tokens, B=8, mode {dt}", errs, bounds[dt], extra=f"model output {e_out:.2e}")
        assert errs[worst] <= bounds[dt], (dt, worst, errs[worst])
        del hooks, out, net
        torch.cuda.empty_cache()
    print(lines[0])
    odir = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
    if os.path.isdir(odir):
        with open(os.path.join(odir, "flux_depth_parity.txt"), "w") as f:
            f.write("\n".join(lines) + "\n")


def test_sdxl_1024_full_layer_set_with_maps_batch1():
    """config_xl_full (every one of the 612 ids incl. the `*-map` hooks: 8.3 GB of hooks per image) on the TRUE SDXL UNet at
    1024^2, batch 1: the eager-processor path (attn_map_kernel at 64-wide heads, 4096^2 and 1024^2 maps) vs the oracle."""
    _threads()
    arch = R.ARCHS["xl"]
    P = R.synth_params(arch, seed=0)
    I = R.synth_inputs(arch, 1, 128, seed=1)
    ids = R.stored_hook_ids(arch)
    assert len(ids) == 612
    maps = [i for i in ids if i.endswith("-map")]
    # the non-map hooks of this configuration are covered (at batch 16) by test_sdxl_1024_batch16_all_non_map_hooks; requesting
    # any map switches every attention layer to the map-materialising kernel, so a sample of non-map hooks rides along
    some = [i for i in ids if not i.endswith("-map")][::13]
    want = [i for i in ids if i in set(maps) | set(some)]
    ref = _oracle(arch, P, I, want, want_map=True)
    u = _native(arch, P)
    g = lambda k: I[k].cuda()
    _, hooks = u.forward_raw(g("sample"), g("timestep"), g("ctx"), g("text_embeds"), g("time_ids"), hook_ids=want)
    torch.cuda.synchronize()
    assert list(hooks.keys()) == want
    errs = {k: max(_rel_each(hooks[k], ref[k])) for k in want}
    ev = sorted(errs.values())
    print(f"\n[sdxl 1024^2 B=1, {len(maps)} maps + {len(some)} other hooks] median {ev[len(ev) // 2]:.2e} worst {ev[-1]:.2e}")
    _check(errs, None, lambda kd: 1.5e-3 if kd == "map" else (1.3e-3 if kd in ("ffn-inner", "unet-out") else 1.0e-3), tag="SDXL 1024^2 B=1 plain plan, 140 maps + riders (maps: 1.5e-3)")
    # ---- PRECISE plan: the 140 maps (and the riders) <= 1e-3 ----
    del hooks, u
    torch.cuda.empty_cache()
    up = _native(arch, P, precise=True)
    _, hooks = up.forward_raw(g("sample"), g("timestep"), g("ctx"), g("text_embeds"), g("time_ids"), hook_ids=want)
    torch.cuda.synchronize()
    errs_p = {k: max(_rel_each(hooks[k], ref[k])) for k in want}
    ev = sorted(errs_p.values())
    print(f"[sdxl 1024^2 B=1 PRECISE, {len(maps)} maps + {len(some)} other hooks] median {ev[len(ev) // 2]:.2e} worst {ev[-1]:.2e}")
    _check(errs_p, None, lambda kd: 1.0e-3, tag="SDXL 1024^2 B=1 PRECISE, 140 maps + riders")
    # ---- the PRODUCT DEFAULT ('auto'): maps select the selective split; the 140 maps (and the riders) <= 1e-3 ----
    del hooks, up
    torch.cuda.empty_cache()
    from components.native import SELECTIVE_BY_ARCH
    ua = _native(arch, P, precise="auto")
    _, hooks = ua.forward_raw(g("sample"), g("timestep"), g("ctx"), g("text_embeds"), g("time_ids"), hook_ids=want)
    torch.cuda.synchronize()
    assert ua.last_split == SELECTIVE_BY_ARCH["xl"]
    errs_a = {k: max(_rel_each(hooks[k], ref[k])) for k in want}
    ev = sorted(errs_a.values())
    print(f"[sdxl 1024^2 B=1 AUTO -> selective split, {len(maps)} maps + {len(some)} other hooks] median {ev[len(ev) // 2]:.2e} worst {ev[-1]:.2e}")
    _check(errs_a, None, lambda kd: 1.0e-3, tag="SDXL 1024^2 B=1 AUTO -> selective split, 140 maps + riders")


def test_pixart_sigma_full_width_batch4():
    """PixArt-Sigma-XL-2 widths (16 heads x 72, caption width 4096, 4096 image + 300 caption tokens, ragged caption mask) with
    a reduced stack (3 of 28 blocks), batch 4 on two samples: every non-map hook vs oracle/pixart_ref.py."""
    _threads()
    from oracle import pixart_ref as PR
    from components.native import NativePixArtTransformer
    arch = dict(PR.ARCH_PIXART_SIGMA); arch.update(num_layers=3)
    P = PR.synth_params(arch, seed=0)
    I = PR.synth_inputs(arch, 2, 128, 300, seed=1, valid=[300, 117])
    st = PR.Store(None)
    with torch.no_grad():
        y = PR.pixart_forward(P, arch, I["hidden_states"], I["encoder_hidden_states"], I["timestep"], I["encoder_attention_mask"], st,
                              want_map=False)
    net = NativePixArtTransformer(arch, device="cuda:0")
    net.load_state_dict({k: v.half() for k, v in P.items()})
    rep = lambda t: torch.cat([t, t], 0).cuda()                       # batch 4 = the two samples twice
    ids = PR.hook_ids(arch)
    out, hooks = net.forward_raw(rep(I["hidden_states"]), rep(I["encoder_hidden_states"]), rep(I["timestep"]),
                                 rep(I["encoder_attention_mask"]), hook_ids=ids)
    torch.cuda.synchronize()
    assert list(hooks.keys()) == ids
    errs = {k: max(_rel_each(hooks[k], st.feats[k])) for k in ids}
    errs["output"] = max(_rel_each(out, y))
    worst = max(errs, key=errs.get)
    print(f"\n[pixart-sigma widths B=4, {len(ids)} hooks] worst {worst} = {errs[worst]:.2e}")
    _margin("PixArt-Sigma widths, 3 blocks, B=4", errs, 6.0e-4)
    assert errs[worst] <= 6.0e-4, (worst, errs[worst])


def test_pixart_sigma_full_depth_batch4():
    """PixArt-Sigma-XL-2-1024 at FULL DEPTH (28 blocks, 16 heads x 72, 4096 image + 300 caption tokens, ragged caption mask), batch 4 on
    two samples: the `out` hook of every block (error vs depth) + the q / ffn-inner hooks of a few blocks + the model output vs
    oracle/pixart_ref.py (reference contract: Transformer2DModel.forward ada_norm_single path, transformer_2d.py:404-475)."""
    _threads()
    from oracle import pixart_ref as PR
    from components.native import NativePixArtTransformer
    arch = dict(PR.ARCH_PIXART_SIGMA)
    nl = arch["num_layers"]
    assert nl == 28
    P = PR.synth_params(arch, seed=0)
    I = PR.synth_inputs(arch, 2, 128, 300, seed=1, valid=[300, 117])
    all_ids = PR.hook_ids(arch)
    ids = [i for i in all_ids if i.endswith("-out") or (i.split("-")[1] in ("block0", "block13", "block27") and i.endswith(("self-q", "cross-q", "ffn-inner")))]
    st = PR.Store({k: True for k in ids})
    with torch.no_grad():
        y = PR.pixart_forward(P, arch, I["hidden_states"], I["encoder_hidden_states"], I["timestep"], I["encoder_attention_mask"], st,
                              want_map=False)
    assert list(st.feats.keys()) == ids
    net = NativePixArtTransformer(arch, device="cuda:0")
    net.load_state_dict({k: v.half() for k, v in P.items()})
    rep = lambda t: torch.cat([t, t], 0).cuda()                       # batch 4 = the two samples twice
    out, hooks = net.forward_raw(rep(I["hidden_states"]), rep(I["encoder_hidden_states"]), rep(I["timestep"]),
                                 rep(I["encoder_attention_mask"]), hook_ids=ids)
    torch.cuda.synchronize()
    assert list(hooks.keys()) == ids
    errs = {k: max(_rel_each(hooks[k], st.feats[k])) for k in ids}
    errs["output"] = max(_rel_each(out, y))
    depth = [errs[f"vit-block{b}-out"] for b in range(nl)]
    worst = max(errs, key=errs.get)
    print(f"\n[pixart-sigma full depth B=4, {len(ids)} hooks] block `out` by depth: " + " ".join(f"{b}:{e:.2e}" for b, e in enumerate(depth)))
    print(f"[pixart-sigma full depth] worst {worst} = {errs[worst]:.2e}; output {errs['output']:.2e}")
    _margin("PixArt-Sigma FULL depth 28 blocks, 4096+300 tokens, B=4", errs, 1.0e-3)
    assert errs[worst] <= 1.0e-3, (worst, errs[worst])


def test_vae_encode_1024_batch2():
    """The step before the hot path at its BASELINE size: SD / SDXL AutoencoderKL encoder (128-256-512-512) on two 1024^2 images,
    posterior sample + Euler noise-add (SDXL scalars) vs oracle/vae_ref.py."""
    _threads()
    from oracle import vae_ref as VR
    from components.native import NativeVAEEncoder
    from helpers import rel_l2
    arch = VR.ARCH_SD_VAE
    P = VR.synth_params(arch, seed=0)
    gen = torch.Generator().manual_seed(1)
    image = (torch.rand(2, 3, 1024, 1024, generator=gen) * 2 - 1).half().float()
    eps = torch.randn(2, 4, 128, 128, generator=gen).half().float()
    noise = torch.randn(2, 4, 128, 128, generator=gen).half().float()
    kw = dict(scaling_factor=0.13025, noise_a=1.0, noise_b=0.7, input_scale=0.82)
    with torch.no_grad():
        ref = VR.prepare_latents(P, arch, image, eps, noise, kw["scaling_factor"], kw["noise_a"], kw["noise_b"], kw["input_scale"])
    enc = NativeVAEEncoder(dict(in_channels=3, latent_channels=4, block_out_channels=arch["block_out_channels"], layers_per_block=2,
                                use_quant_conv=1), device="cuda:0")
    enc.load_vae_state_dict({k: v.half() for k, v in P.items()})
    got = enc.encode(image, eps=eps, noise=noise, **kw)
    torch.cuda.synchronize()
    e = rel_l2(got, ref)
    print(f"\n[vae 1024^2 B=2] rel L2 {e:.2e}")
    _margin("VAE encode + sample + noise-add 1024^2 B=2 (latents)", {"latents": e}, 1e-3)
    assert got.shape == ref.shape == (2, 4, 128, 128) and e <= 1e-3, e


def test_vae_out_decode_1024_batch2():
    """`vae-out` at its BASELINE size (reference diffusion_feature.py:477-485): one Euler step on 128x128 latents + the SD / SDXL
    AutoencoderKL decoder (512-512-256-128, 49.5 M parameters, 10.5 TFLOP per 1024^2 image) on two images (one sample repeated) vs
    oracle/vae_ref.py at batch 1."""
    _threads()
    from oracle import vae_ref as VR
    from components.native import NativeVAEDecoder
    from helpers import rel_l2
    arch = VR.ARCH_SD_VAE
    P = VR.synth_dec_params(arch, seed=0)
    assert sum(v.numel() for v in P.values()) == 49490199          # decoder + post_quant_conv of the published 83,653,863-parameter VAE
    gen = torch.Generator().manual_seed(1)
    lat = torch.randn(1, 4, 128, 128, generator=gen).half().float()
    eps = torch.randn(1, 4, 128, 128, generator=gen).half().float()
    a, b, sf = 1.0, -0.4, 0.13025
    with torch.no_grad():
        ref = VR.vae_out(P, arch, lat, eps, a, b, sf)
    dec = NativeVAEDecoder(dict(in_channels=3, latent_channels=4, block_out_channels=arch["block_out_channels"], layers_per_block=2,
                                use_quant_conv=1), device="cuda:0")
    dec.load_vae_state_dict({k: v.half() for k, v in P.items()})
    got = dec.decode(lat.expand(2, -1, -1, -1).contiguous(), eps.expand(2, -1, -1, -1).contiguous(), c_sample=a, c_eps=b, scaling_factor=sf)
    torch.cuda.synchronize()
    assert tuple(got.shape) == (2, 3, 1024, 1024)
    e = max(rel_l2(got[i:i + 1], ref) for i in range(2))
    print(f"\n[vae-out 1024^2 B=2] rel L2 {e:.2e}")
    _margin("'vae-out': scheduler step + VAE decode 1024^2 B=2", {"vae-out": e}, 1e-3)
    assert e <= 1e-3, e


@pytest.mark.parametrize("ver,lat,B", [("xl", 56, 3), ("1-5", 40, 5)])
def test_true_width_ragged_shapes_odd_batch(ver, lat, B):
    """Robustness beyond the BASELINE shapes: the TRUE SDXL / SD1.5 widths on a latent grid whose row counts are not multiples of the 256- / 128-row
    tiles (SDXL 448^2: 3136 / 784 / 196 tokens per image; SD1.5 320^2: 1600 / 400 / 100 / 25), an ODD batch of DIFFERENT samples and prompts, attention
    sequences that are not multiples of the 64-key tile: ragged last tiles in every GEMM / conv, masked tail tiles in every attention, both GroupNorm
    paths.  (Square grids: the reference's FeatureStore reshapes token hooks to sqrt(tokens)^2, feature_extractor.py:46-48.)  Every 5th non-map hook
    (+ `unet-out`) on the automatically chosen plan, every sample against the oracle: every kind <= 1e-3 (the chooser's contract), plus the plain plan
    within 1.45e-3 on `ffn-inner` / `unet-out`, 1.1e-3 elsewhere."""
    _threads()
    arch = R.ARCHS[ver]
    P = R.synth_params(arch, seed=0)
    I = R.synth_inputs(arch, B, lat, seed=3, same_prompt=False)
    h = w = lat
    allids = [i for i in R.stored_hook_ids(arch) if not i.endswith("-map")]
    ids = [i for n, i in enumerate(allids) if n % 5 == 0 or i == "unet-out"]
    ref = _oracle(arch, P, I, ids)
    gi = lambda k: I[k].cuda() if k in I else None
    # (plain plan: the fp16-operand floor itself is a little higher on smaller grids — fewer elements to average over: 1.39e-3 on `unet-out` at 448^2)
    for spec, bound in (("auto", lambda kd: 1.0e-3), (False, lambda kd: 1.45e-3 if kd in ("ffn-inner", "unet-out") else 1.1e-3)):
        u = _native(arch, P, precise=spec)
        _, hooks = u.forward_raw(gi("sample"), gi("timestep"), gi("ctx"), gi("text_embeds"), gi("time_ids"), hook_ids=ids)
        torch.cuda.synchronize()
        assert list(hooks.keys()) == ids
        errs = {}
        for k in ids:
            assert tuple(hooks[k].shape) == tuple(ref[k].shape), (k, hooks[k].shape, ref[k].shape)
            assert torch.isfinite(hooks[k].float()).all(), k
            errs[k] = max(_rel_each(hooks[k], ref[k]))
        ev = sorted(errs.values())
        print(f"\n[{ver} {8 * h}x{8 * w} B={B} plan {spec} -> mask {u.last_split}] hooks={len(ev)} median {ev[len(ev) // 2]:.2e} worst {ev[-1]:.2e}")
        _check(errs, None, bound)
        del hooks, u
        torch.cuda.empty_cache()


def test_sd21_512_plan_levels_borrow_the_sd15_table():
    """`2-1` (SD2.1-base: the SD1.5 topology with 64-wide heads, linear projections, cross dim 1024) is served by the SD1.5 family's operand-error
    table (components/native.py arch_family).  That is an assumption about another model: checked here at true widths, 512^2, B = 2 (two samples,
    two prompts) — every hook the chooser hands to the plain / light level alone must measure <= 0.97e-3 under that level, and the
    automatically chosen plan of the full non-map layer set must keep every kind within 1e-3."""
    _threads()
    from components.native import SPLIT_LIGHT, SELECTIVE_BY_ARCH
    arch = R.ARCHS["2-1"]
    P = R.synth_params(arch, seed=0)
    I = R.synth_inputs(arch, 2, 64, seed=1, same_prompt=False)
    ids = [i for i in R.stored_hook_ids(arch) if not i.endswith("-map")]
    ref = _oracle(arch, P, I, ids)
    g = lambda k: I[k].cuda()
    run = lambda u: u.forward_raw(g("sample"), g("timestep"), g("ctx"), hook_ids=ids)[1]
    u = _native(arch, P)
    hooks = run(u)
    torch.cuda.synchronize()
    errs = {k: max(_rel_each(hooks[k], ref[k])) for k in ids}
    ev = sorted(errs.values())
    print(f"\n[sd2.1 512^2 B=2 plain, {len(ids)} ids] median {ev[len(ev) // 2]:.2e} worst {ev[-1]:.2e}; below 1e-3: {sum(e < 1e-3 for e in ev)}")
    del hooks, u
    torch.cuda.empty_cache()
    _plain_plan_contract("1-5", arch, errs, "sd21_b2")
    errs_l = _light_level_errs(arch, P, run, ref, ids)
    _plain_plan_contract("1-5", arch, errs_l, "sd21_b2", SPLIT_LIGHT)
    ua = _native(arch, P, precise="auto")
    hooks = run(ua)
    torch.cuda.synchronize()
    assert ua.last_split == SELECTIVE_BY_ARCH["1-5"]
    errs_a = {k: max(_rel_each(hooks[k], ref[k])) for k in ids}
    ev = sorted(errs_a.values())
    print(f"[sd2.1 512^2 B=2 AUTO -> selective split] median {ev[len(ev) // 2]:.2e} worst {ev[-1]:.2e}")
    _check(errs_a, None, lambda kd: 1.0e-3)


def test_sdxl_plan_level_contract_on_other_inputs():
    """The operand-error table was emulated on ONE seeded sample / prompt at t = 100.  The per-hook contract of the plain and light levels
    (_plain_plan_contract) on DIFFERENT inputs: two other samples with two other prompts at t = 500, true SDXL 1024^2, B = 2."""
    _threads()
    from components.native import SPLIT_LIGHT
    arch = R.ARCHS["xl"]
    P = R.synth_params(arch, seed=0)
    I = R.synth_inputs(arch, 2, 128, seed=23, same_prompt=False)
    I["timestep"] = torch.tensor([500.0])
    ids = [i for i in R.stored_hook_ids(arch) if not i.endswith("-map")]
    ref = _oracle(arch, P, I, ids)
    g = lambda k: I[k].cuda()
    run = lambda u: u.forward_raw(g("sample"), g("timestep"), g("ctx"), g("text_embeds"), g("time_ids"), hook_ids=ids)[1]
    u = _native(arch, P)
    hooks = run(u)
    torch.cuda.synchronize()
    errs = {k: max(_rel_each(hooks[k], ref[k])) for k in ids}
    ev = sorted(errs.values())
    print(f"\n[sdxl 1024^2 B=2, other samples / prompts, t=500, plain] median {ev[len(ev) // 2]:.2e} worst {ev[-1]:.2e}; below 1e-3: {sum(e < 1e-3 for e in ev)}")
    del hooks, u
    torch.cuda.empty_cache()
    _plain_plan_contract("xl", arch, errs, "sdxl_b2_other_inputs", product=False)
    errs_l = _light_level_errs(arch, P, run, ref, ids)
    _plain_plan_contract("xl", arch, errs_l, "sdxl_b2_other_inputs", SPLIT_LIGHT, product=False)


def test_sdxl_heavy_tailed_full_split_is_a_reference():
    """Weights that are NOT N(0, 1/fan_in) (oracle/unet_ref.py synth_params_heavy: log-normal per-channel scales, x8 outlier channels in the residual
    stream — the draw on which, before the q / k / v pairs, NO plan level was inside 1e-3: full split 1.50e-3, profiles/r05_heavy_tailed_plan_levels.txt).
    The full split — every GEMM / conv operand class AND, since round 5, the q / k / v of both attentions as pairs (csrc/attn.hip QKP; the storage
    rounding in front of the text cross-attention's peaked softmax was 8e-4 of that floor by itself) — stays near the hook-storage floor: every
    non-map hook <= 4e-4, while the plain plan is beyond 1.5e-3.  This is what makes `verify`'s yardstick (the distance to the full split) a
    statement about fp32 on such statistics.  True SDXL widths, 1024^2, batch 1."""
    _threads()
    arch = R.ARCHS["xl"]
    P = R.synth_params_heavy(arch, seed=0, outlier_gain=8.0)
    I = R.synth_inputs(arch, 1, 128, seed=1)
    ids = [i for i in R.stored_hook_ids(arch) if not i.endswith("-map")]
    ref = _oracle(arch, P, I, ids)
    assert all(torch.isfinite(v).all() for v in ref.values())
    g = lambda k: I[k].cuda()
    worst = {}
    for name, precise in (("plain", False), ("full split", True)):
        u = _native(arch, P, precise=precise)
        hooks = u.forward_raw(g("sample"), g("timestep"), g("ctx"), g("text_embeds"), g("time_ids"), hook_ids=ids)[1]
        torch.cuda.synchronize()
        errs = {k: max(_rel_each(hooks[k], ref[k])) for k in ids}
        ev = sorted(errs.values())
        worst[name] = ev[-1]
        print(f"\n[sdxl 1024^2 B=1 heavy-tailed x8, {name}] median {ev[len(ev) // 2]:.2e} worst {ev[-1]:.2e} ({max(errs, key=errs.get)})")
        del hooks, u
        torch.cuda.empty_cache()
    assert worst["full split"] <= 4e-4 and worst["plain"] > 1.5e-3, worst
    # ---- and the product path on these weights: precise='auto' + verify.  The table (built on benign statistics) hands this layer set to the
    # selective preset, which is NOT enough here; the ladder measures that against the full split and keeps the first level that is within
    # 9.5e-4 of it — the DEEP level (selective + q / k / v pairs + the GEGLU operand), not the full split — and what it hands out is inside
    # 1e-3 of the fp32 oracle on every hook
    import warnings
    from components.native import NativeUNet, SPLIT_ALL, SELECTIVE_BY_ARCH, SPLIT_DEEP_EXTRA
    u = NativeUNet(cfg_from_oracle_arch(arch), device="cuda:0", precise="auto", verify=True)
    u.load_state_dict({k: v.half() for k, v in P.items()})
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        hooks = u.forward_raw(g("sample"), g("timestep"), g("ctx"), g("text_embeds"), g("time_ids"), hook_ids=ids)[1]
        torch.cuda.synchronize()
    key, seen, kept = u.verify_log[0]
    errs = {k: max(_rel_each(hooks[k], ref[k])) for k in ids}
    print(f"[sdxl heavy-tailed x8, auto + verify] distances to the full split { {m: '%.2e' % v for m, v in seen.items()} } -> kept {kept}: worst {max(errs.values()):.2e} vs fp32")
    assert len([x for x in w if "gdf verify" in str(x.message)]) == 1
    from components.native import SPLIT_CLASSES
    assert kept in (SELECTIVE_BY_ARCH["xl"] | SPLIT_CLASSES["ln_ff"], SELECTIVE_BY_ARCH["xl"] | SPLIT_DEEP_EXTRA) and seen[SELECTIVE_BY_ARCH["xl"]] > u.verify_accept_bound()
    assert seen[kept] ** 2 + 2.7e-4 ** 2 <= 0.97e-3 ** 2 * (1 + 1e-9)          # round 6: d^2 + e_full^2 <= target^2, not a constant bound on d
    assert max(errs.values()) <= 0.97e-3, max(errs, key=errs.get)
    assert abs(seen[kept] - max(errs.values())) < 1.5e-4          # the yardstick (distance to the full split) tracks the distance to fp32
    _margin("SDXL 1024^2 B=1 HEAVY-TAILED x8 weights, auto + verify -> kept level %d" % kept, errs, 0.97e-3, extra=f"d to full split {seen[kept]:.2e}; full split itself {worst['full split']:.2e}")


@pytest.mark.parametrize("ver,gain", [("xl", 16.0), ("xl", 32.0), ("1-5", 16.0)])
def test_verify_ladder_on_heavy_tailed_weights(ver, gain):
    """`verify` on heavy-tailed weight statistics at TRUE size, other outlier gains and the SD1.5 family.  Asserted: the level the
    ladder keeps is within 0.97e-3 of the fp32 oracle on EVERY non-map hook (the acceptance rule d^2 + e_full^2 <= (0.97e-3)^2 is arithmetic, not a
    constant), and on SD1.5 — whose heavy-tailed draw failed every rung of the round-5 ladder (1.1-1.5e-3) and fell to the full split — a rung
    CHEAPER than the full split is kept."""
    import warnings
    from components.native import NativeUNet, SPLIT_ALL
    _threads()
    arch = R.ARCHS[ver]
    lat = 128 if ver == "xl" else 64
    P = R.synth_params_heavy(arch, seed=0, outlier_gain=gain)
    I = R.synth_inputs(arch, 1, lat, seed=1)
    ids = [i for i in R.stored_hook_ids(arch) if not i.endswith("-map")]
    ref = _oracle(arch, P, I, ids)
    g = lambda k: I[k].cuda() if k in I else None
    u = NativeUNet(cfg_from_oracle_arch(arch), device="cuda:0", precise="auto", verify=True)
    u.load_state_dict({k: v.half() for k, v in P.items()})
    with warnings.catch_warnings(record=True):
        warnings.simplefilter("always")
        hooks = u.forward_raw(g("sample"), g("timestep"), g("ctx"), g("text_embeds"), g("time_ids"), hook_ids=ids)[1]
        torch.cuda.synchronize()
    key, seen, kept = u.verify_log[0]
    errs = {k: max(_rel_each(hooks[k], ref[k])) for k in ids}
    worst = max(errs, key=errs.get)
    print(f"\n[{ver} heavy-tailed x{gain:g}, auto + verify] bound on d {u.verify_accept_bound():.3e}; distances to the full split "
          f"{ {m: '%.2e' % v for m, v in seen.items()} } -> kept {kept}: worst {errs[worst]:.2e} ({worst}) vs fp32")
    _margin(f"{'SDXL 1024^2' if ver == 'xl' else 'SD1.5 512^2'} B=1 HEAVY-TAILED x{gain:g} weights, auto + verify -> kept level {kept}", errs, 0.97e-3,
            extra="levels tried " + " ".join(f"{m}:{v:.2e}" for m, v in seen.items()))
    assert errs[worst] <= 0.97e-3, (worst, errs[worst])
    if kept != SPLIT_ALL:
        e_full = u.FULL_SPLIT_ERROR["xl" if ver == "xl" else "1-5"]
        assert seen[kept] ** 2 + e_full ** 2 <= 0.97e-3 ** 2 * (1 + 1e-9)
    if ver == "1-5":
        assert kept != SPLIT_ALL, seen
    assert {k[8] for k in u._plans} == {kept}                     # the check-only plans (full split, rejected rungs) are gone
