"""Whole-path parity (-m gpu): libgdf.so (through the C ABI, via components.native.NativeUNet) against the CPU
oracle on identical seeded weights / latents / timestep / prompt-embeds.

Stated tolerance for these SHRUNKEN-width models: per hooked tensor  ||gpu - oracle||_2 / ||oracle||_2  <= 2e-3 and
<= 1.3x the fp16-operand floor of the same model (oracle/operand_floor.py): with 64..640 channels a tensor averages the
fp16 operand rounding over far fewer elements than the true widths, so the floor itself reaches 1.5e-3 here.  The north-star
bound (1e-3) is asserted on the TRUE widths at the BASELINE batch sizes in tests/test_gpu_fullsize.py."""
import os

import pytest
import torch

from helpers import cfg_from_oracle_arch, oracle_run, rel_l2
from oracle import unet_ref as R
from oracle.operand_floor import fp16_operands

pytestmark = pytest.mark.gpu
TOL = 2e-3


def native(arch, P, **kw):
    from components.native import NativeUNet
    u = NativeUNet(cfg_from_oracle_arch(arch), device="cuda:0", **kw)
    u.load_state_dict({k: v.half() for k, v in P.items()})
    return u


def run_native(u, I, ids):
    g = lambda k: I[k].cuda() if k in I else None
    noise, hooks = u.forward_raw(g("sample"), g("timestep"), g("ctx"), g("text_embeds"), g("time_ids"), hook_ids=ids)
    torch.cuda.synchronize()
    return noise, hooks


@pytest.mark.parametrize("base,lat,batch", [("xl", 16, 2), ("1-5", 16, 1), ("xl", 32, 1), ("2-1", 16, 2)])
def test_all_hooks_match_oracle(base, lat, batch):
    arch = R.tiny_arch(base)
    P = R.synth_params(arch, seed=0)
    I = R.synth_inputs(arch, batch, lat, seed=1)
    ref = oracle_run(arch, P, I)                                   # accept-all (incl. '-map' hooks)
    u = native(arch, P)
    assert u.hook_names() == R.stored_hook_ids(arch)               # same ids, same execution order
    noise, hooks = run_native(u, I, list(ref.keys()))
    assert list(hooks.keys()) == list(ref.keys())
    errs = {}
    for k, r in ref.items():
        assert tuple(hooks[k].shape) == tuple(r.shape), k
        assert hooks[k].dtype == torch.float16
        errs[k] = rel_l2(hooks[k], r)
    worst = max(errs, key=errs.get)
    print(f"[{base} lat{lat}] hooks={len(errs)} worst {worst} = {errs[worst]:.2e}; median {sorted(errs.values())[len(errs)//2]:.2e}")
    bad = {k: v for k, v in errs.items() if not v < TOL}
    assert not bad, sorted(bad.items(), key=lambda kv: -kv[1])[:8]
    assert rel_l2(noise, ref["unet-out"]) < TOL
    with fp16_operands():
        flo = oracle_run(arch, P, I)
    over = {k: (errs[k], rel_l2(flo[k], r)) for k, r in ref.items() if not errs[k] <= 1.3 * rel_l2(flo[k], r) + 5e-5}
    assert not over, sorted(over.items(), key=lambda kv: -kv[1][0])[:8]


@pytest.mark.parametrize("base,lat,batch", [("xl", 16, 2), ("1-5", 16, 1), ("2-1", 32, 1)])
def test_precise_plan_split_operands_all_hooks(base, lat, batch):
    """Opt-in `precise` plans (include/gdf.h gdf_plan_opts.reserved[1]): every GEMM / conv activation operand and every GroupNorm input
    is a split fp16 pair hi + lo, contracted as [hi | lo] x [W | W].  What remains against the fp32 oracle is the fp16 storage of
    q / k / v / P inside attention and of the hooks themselves: all hooks (incl. maps) of the shrunken models — whose default-plan
    error reaches 1.5e-3 — come in under 6e-4, and the default plan of the same model is measurably worse on the same inputs."""
    arch = R.tiny_arch(base)
    P = R.synth_params(arch, seed=0)
    I = R.synth_inputs(arch, batch, lat, seed=1)
    ref = oracle_run(arch, P, I)
    ids = list(ref.keys())
    up = native(arch, P, precise=True)
    noise, hooks = run_native(up, I, ids)
    assert list(hooks.keys()) == ids
    errs = {k: rel_l2(hooks[k], ref[k]) for k in ids}
    worst = max(errs, key=errs.get)
    _, dflt = run_native(native(arch, P, precise=False), I, ids)
    errs_d = {k: rel_l2(dflt[k], ref[k]) for k in ids}
    md, mp = sorted(errs_d.values())[len(ids) // 2], sorted(errs.values())[len(ids) // 2]
    print(f"[{base} lat{lat} precise] hooks={len(ids)} worst {worst} = {errs[worst]:.2e}; median {mp:.2e} (default plan: worst "
          f"{max(errs_d.values()):.2e}, median {md:.2e})")
    bad = {k: v for k, v in errs.items() if not v < 6e-4}
    assert not bad, sorted(bad.items(), key=lambda kv: -kv[1])[:8]
    assert rel_l2(noise, ref["unet-out"]) < 6e-4
    assert mp < 0.6 * md
    # selected hooks (no maps: the flash attention kernel instead of the map-materialising one) with and without early exit
    sel = [i for i in ids if i.endswith(("ffn-inner", "cross-q", "res-increment", "vit-out"))][:6]
    _, hs = run_native(up, I, sel)
    ue = native(arch, P, precise=True, early_exit=True)
    _, he = run_native(ue, I, sel)
    for k in sel:
        assert torch.equal(he[k], hs[k]), k
        assert rel_l2(hs[k], ref[k]) < 6e-4, k


def test_fp16_stream_option_and_selected_hooks():
    arch = R.tiny_arch("xl")
    P = R.synth_params(arch, seed=3)
    I = R.synth_inputs(arch, 2, 16, seed=4)
    ids = ["up-level0-repeat0-vit-block1-out", "up-level1-repeat0-vit-block0-cross-q", "up-level1-repeat0-vit-block0-out",
           "not-a-layer"]                                           # unknown ids are silently ignored
    ref = oracle_run(arch, P, I, ids)
    for fp32 in (True, False):
        u = native(arch, P, stream_fp32=fp32)
        _, hooks = run_native(u, I, ids)
        assert list(hooks.keys()) == list(ref.keys())
        for k in ref:
            assert rel_l2(hooks[k], ref[k]) < (TOL if fp32 else 6e-3), (k, fp32)


def test_early_exit_matches_full_run():
    arch = R.tiny_arch("xl")
    P = R.synth_params(arch, seed=0)
    I = R.synth_inputs(arch, 1, 16, seed=1)
    ids = ["down-level1-repeat1-vit-block1-ffn-inner", "mid-vit-block0-self-k"]
    u = native(arch, P)
    _, full = run_native(u, I, ids)
    u2 = native(arch, P, early_exit=True)
    _, ee = run_native(u2, I, ids)
    for k in ids:
        assert torch.equal(full[k], ee[k])                          # same kernels, same inputs: bit-identical


def test_shared_ctx_plan_equals_per_sample_ctx():
    """shared_ctx (one prompt repeated, reference diffusion_feature.py:272) computes text K/V once: same numbers."""
    arch = R.tiny_arch("xl")
    P = R.synth_params(arch, seed=0)
    I = R.synth_inputs(arch, 3, 16, seed=1, same_prompt=True)
    ids = ["down-level1-repeat0-vit-block0-cross-q", "mid-vit-block1-out", "up-level1-repeat2-vit-out"]
    u = native(arch, P)
    g = lambda k: I[k].cuda()
    _, a = u.forward_raw(g("sample"), g("timestep"), g("ctx"), g("text_embeds"), g("time_ids"), hook_ids=ids)
    _, b = u.forward_raw(g("sample"), g("timestep"), g("ctx"), g("text_embeds"), g("time_ids"), hook_ids=ids, shared_ctx=True)
    torch.cuda.synchronize()
    for k in ids:
        assert torch.equal(a[k], b[k]), k


def test_determinism_and_fresh_outputs():
    arch = R.tiny_arch("1-5")
    P = R.synth_params(arch, seed=0)
    I = R.synth_inputs(arch, 1, 16, seed=1)
    ids = ["up-level1-repeat2-res-out", "up-level3-repeat0-vit-block0-self-k"]
    u = native(arch, P)
    _, a = run_native(u, I, ids)
    _, b = run_native(u, I, ids)
    for k in ids:
        assert torch.equal(a[k], b[k]) and a[k].data_ptr() != b[k].data_ptr()


def test_feature_extractor_api_synthetic(tmp_path, monkeypatch):
    """README.md:82-113 usage through the drop-in class, on the synthetic front-end (no checkpoints offline)."""
    import json
    from PIL import Image
    import numpy as np
    monkeypatch.setenv("GDF_SYNTHETIC_WEIGHTS", "1")
    import diffusion_feature
    cfg = tmp_path / "layers.json"
    cfg.write_text(json.dumps({"up-level1-repeat1-vit-block0-cross-q": True, "up-level1-repeat2-res-out": True,
                               "up-level2-repeat1-vit-block0-cross-q": True, "up-level3-repeat0-vit-block0-self-k": True}))
    df = diffusion_feature.FeatureExtractor(layer=str(cfg), version='1-5', img_size=256, device='cuda')
    prompt = df.encode_prompt('a photo of a cat')
    img = Image.fromarray((np.random.RandomState(0).rand(300, 280, 3) * 255).astype(np.uint8))
    feats = df.extract(prompt, batch_size=2, image=[img, img], t=100)
    assert list(feats.keys()) == ["up-level1-repeat1-vit-block0-cross-q", "up-level1-repeat2-res-out",
                                  "up-level2-repeat1-vit-block0-cross-q", "up-level3-repeat0-vit-block0-self-k"]
    shapes = {k: tuple(v.shape) for k, v in feats.items()}
    assert shapes == {"up-level1-repeat1-vit-block0-cross-q": (2, 1280, 8, 8), "up-level1-repeat2-res-out": (2, 1280, 8, 8),
                      "up-level2-repeat1-vit-block0-cross-q": (2, 640, 16, 16),
                      "up-level3-repeat0-vit-block0-self-k": (2, 320, 32, 32)}
    assert sum(s[1] for s in shapes.values()) == 3520               # correspondence config `feature_len`
    for v in feats.values():
        assert v.dtype == torch.float16 and v.is_cuda and torch.isfinite(v.float()).all()


def test_aggregated_attention_feature(monkeypatch):
    """`attention=[...]` (reference diffusion_feature.py:67-68, 492-500): feats['attn'] = head-mean maps grouped by size."""
    monkeypatch.setenv("GDF_SYNTHETIC_WEIGHTS", "1")
    import diffusion_feature
    from components.feature_extractor import aggregate_attention
    layer = {"up-level1-repeat0-vit-block0-cross-map": True, "up-level2-repeat2-vit-block0-cross-map": True}
    df = diffusion_feature.FeatureExtractor(layer=layer, version='1-5', img_size=256, device='cuda', attention=['up_cross'])
    prompt = df.encode_prompt('a photo of a cat')
    lat = torch.randn(2, 4, 32, 32, generator=torch.Generator().manual_seed(0)).half()
    feats = df.extract(prompt, batch_size=2, image=lat, image_type='latents', t=100)
    assert list(feats.keys()) == list(layer.keys()) + ['attn']
    # grids in [256/32, 256/16] = [8, 16]: up-level1 (8x8) and up-level2 (16x16), 77 text tokens each
    assert feats['attn'].shape == (2, 154, 32, 32)
    # the two explicitly requested maps are members of the two size groups; each group averages 3 layers, so only check
    # normalisation: every map row sums to 1 over the 77 keys -> channel sum == 1 everywhere
    s = feats['attn'].float()
    assert torch.allclose(s[:, :77].sum(1), torch.ones(2, 32, 32, device=s.device), atol=5e-3)
    assert torch.allclose(s[:, 77:].sum(1), torch.ones(2, 32, 32, device=s.device), atol=5e-3)


def test_hipgraph_replay_equals_eager():
    """Product default (gdf_plan_set_graph + components.native._Plan): every forward runs on the plan's private stream over
    buffers with stable addresses; after one eager warm-up forward the op program is captured ONCE per hook-buffer set and
    replayed.  A set is handed out again only when the caller dropped every tensor of it, so results stay valid."""
    from components.native import NativeUNet
    arch = R.tiny_arch("xl")
    P = R.synth_params(arch, seed=0)
    I = R.synth_inputs(arch, 2, 16, seed=1)
    unet = NativeUNet(cfg_from_oracle_arch(arch), device="cuda:0")
    unet.load_state_dict({k: v.half() for k, v in P.items()})
    ids = ["down-level1-repeat0-vit-block0-cross-q", "mid-vit-block1-out", "up-level2-repeat2-res-out", "unet-out"]
    args = [I[k].cuda() for k in ("sample", "timestep", "ctx", "text_embeds", "time_ids")]
    _, eager = unet.forward_raw(*args, hook_ids=ids)                 # first forward of a plan: eager (lazy kernel attribute setup)
    torch.cuda.synchronize()
    plan = unet._plan(2, 16, 16, 77, ids, False)
    assert plan.graph_stats() == (0, 0, 0)
    outs = [unet.forward_raw(*args, hook_ids=ids)[1] for _ in range(4)]   # all results kept alive: two more pooled sets get a
    torch.cuda.synchronize()                                              # graph each, the rest run eagerly on one-off buffers
    assert plan.graph_stats() == (2, 2, 0), plan.graph_stats()
    ptrs = {o[ids[0]].data_ptr() for o in outs} | {eager[ids[0]].data_ptr()}
    assert len(ptrs) == 5                                                 # five live results, five different buffers
    for o in outs:
        for k in ids:
            assert torch.equal(o[k], eager[k]), k
    # steady state: results dropped before the next call -> the same set is reused -> replays only, no capture
    del outs, o
    for _ in range(6):
        last = unet.forward_raw(*args, hook_ids=ids)[1]
        torch.cuda.synchronize()
        chk = {k: v.clone() for k, v in last.items()}
        del last
    assert plan.graph_stats() == (2, 8, 0), plan.graph_stats()
    for k in ids:
        assert torch.equal(chk[k], eager[k]), k
    # new input VALUES flow through the staging buffers of the captured graph
    args2 = [a.clone() for a in args]
    args2[0] = torch.randn_like(args2[0])
    _, other = unet.forward_raw(*args2, hook_ids=ids)
    ref2 = oracle_run(arch, P, dict(I, sample=args2[0].float().cpu()), ids)
    assert plan.graph_stats()[0] == 2
    for k in ids:
        assert not torch.equal(other[k], eager[k]) and rel_l2(other[k], ref2[k]) < TOL, k


def test_feature_extractor_steady_state_replays_one_graph(monkeypatch):
    """FeatureExtractor.extract in a loop (results consumed, then dropped): no capture after the start-up phase, host side is a
    single hipGraphLaunch per forward, features identical call to call."""
    monkeypatch.setenv("GDF_SYNTHETIC_WEIGHTS", "1")
    import diffusion_feature
    layer = {"up-level1-repeat1-vit-block0-cross-q": True, "up-level1-repeat2-res-out": True}
    df = diffusion_feature.FeatureExtractor(layer=layer, version='1-5', img_size=256, device='cuda')
    prompt = df.encode_prompt('a photo of a cat')
    lat = torch.randn(2, 4, 32, 32, generator=torch.Generator().manual_seed(0)).half()
    first = {k: v.clone() for k, v in df.extract(prompt, batch_size=2, image=lat, image_type='latents', t=100).items()}
    feats = None
    for _ in range(8):
        feats = df.extract(prompt, batch_size=2, image=lat, image_type='latents', t=100)   # previous dict alive during the call
    plan = next(iter(df.pipe.unet._plans.values()))
    cap, lau, _fails = plan.graph_stats()
    assert cap <= 2 and lau == 8, (cap, lau)
    for k in first:
        assert torch.equal(first[k], feats[k])


@pytest.mark.parametrize("tag", ["xl", "15", "21"])
def test_matches_reference_whole_unet_golden(tag):
    """HIP path vs tests/golden/unet_tiny_*.npz = the reference's own UNet2DConditionModel.forward (gen_golden_unet.py):
    relative L2 error over the 2048 sampled positions of every hook (the full tensors are covered by the oracle tests)."""
    from test_oracle_golden import unet_golden
    meta, I, gold, out = unet_golden(tag)
    arch = meta["arch"]
    P = R.synth_params(arch, seed=meta["wseed"])
    u = native(arch, P)
    noise, hooks = run_native(u, I, meta["order"])
    assert list(hooks.keys()) == meta["order"]
    worst = ("", 0.0)
    for k, (idx, vals, norm, shape) in gold.items():
        got = hooks[k].float().cpu().contiguous()
        assert tuple(got.shape) == shape, k
        e = float((got.flatten()[idx] - vals).norm() / (vals.norm() + 1e-30))
        worst = max(worst, (k, e), key=lambda t: t[1])
        assert e < 2.5e-3, (k, e)
    assert rel_l2(noise, out) < TOL
    print(f"[golden {tag}] worst sampled rel L2 {worst[1]:.2e} at {worst[0]}")


def test_aggregated_attention_feature_matches_oracle():
    """(f)3 end to end: FeatureExtractor(attention=[...]).extract -> feats['attn'] on the HIP path (maps written by
    attn_map_kernel) against the oracle's aggregation (oracle/attn_agg_ref.py, pinned to the reference AttentionStore) of the
    ORACLE's own attention maps for the same weights / latents / timestep / prompt embeds."""
    import diffusion_feature
    from components.models import SyntheticPipe
    from components.native import NativeUNet
    from oracle import attn_agg_ref as AR
    arch = R.ARCHS["1-5"]
    P = R.synth_params(arch, seed=0)
    pipe = SyntheticPipe("1-5", "cuda:0", seed=0)
    pipe.unet.load_state_dict({k: v.half() for k, v in P.items()})
    sel = ["up_cross", "down_self"]
    layer = {"up-level1-repeat0-vit-block0-cross-map": True, "up-level2-repeat2-vit-block0-out": True}
    df = diffusion_feature.FeatureExtractor(layer=layer, version='1-5', img_size=256, device='cuda:0', attention=sel,
                                            external_model=pipe)
    seen = {}
    raw = NativeUNet.forward_raw

    def spy(self, sample, timestep, encoder_hidden_states, *a, **kw):
        seen.update(sample=sample.float().cpu(), timestep=torch.as_tensor(timestep).float().cpu().reshape(-1)[:1],
                    ctx=encoder_hidden_states.float().cpu())
        return raw(self, sample, timestep, encoder_hidden_states, *a, **kw)
    NativeUNet.forward_raw = spy
    try:
        prompt = df.encode_prompt('a photo of a cat')
        lat = torch.randn(2, 4, 32, 32, generator=torch.Generator().manual_seed(0)).half()
        feats = df.extract(prompt, batch_size=2, image=lat, image_type='latents', t=100)
    finally:
        NativeUNet.forward_raw = raw
    assert list(feats.keys()) == list(layer.keys()) + ['attn']
    st = R.Store(None)                                                # accept-all: every map, like the reference's AttentionStore sees
    with torch.no_grad():
        R.unet_forward(P, arch, seen["sample"], seen["timestep"], seen["ctx"], store=st, want_map=True)
    maps = [(k, v) for k, v in st.feats.items() if k.endswith("-map")]
    want = AR.aggregate(maps, sel, 256 // 32, 256 // 16, 256 // 8)
    got = feats['attn']
    assert got.shape == want.shape == (2, 77 + 77 + 256 + 64, 32, 32)        # up_cross @ 8x8, 16x16; down_self @ 16x16 (K = 256), 8x8 (K = 64)
    e = rel_l2(got, want)
    print(f"[attn aggregate] shape {tuple(got.shape)} rel L2 {e:.2e}")
    assert e < 1e-3, e
    for k in layer:
        assert rel_l2(feats[k], st.feats[k]) < TOL, k


def test_feature_extractor_precise_env(monkeypatch):
    """GDF_PRECISE=1 switches an unmodified FeatureExtractor caller to split-operand plans: same ids, same shapes,
    values within the fp16-operand distance of the default plans'."""
    monkeypatch.setenv("GDF_SYNTHETIC_WEIGHTS", "1")
    import diffusion_feature
    layer = {"up-level1-repeat1-vit-block0-cross-q": True, "up-level1-repeat2-res-out": True, "up-level2-repeat1-vit-block0-ffn-inner": True}
    lat = torch.randn(2, 4, 32, 32, generator=torch.Generator().manual_seed(0)).half()
    out = {}
    for tag, val in (("default", "0"), ("precise", "1")):
        monkeypatch.setenv("GDF_PRECISE", val)
        df = diffusion_feature.FeatureExtractor(layer=layer, version='1-5', img_size=256, device='cuda')
        assert df.pipe.unet.precise == (val == "1")
        prompt = df.encode_prompt('a photo of a cat')
        out[tag] = {k: v.clone() for k, v in df.extract(prompt, batch_size=2, image=lat, image_type='latents', t=100).items()}
    assert list(out["default"].keys()) == list(out["precise"].keys())
    for k in out["default"]:
        a, b = out["default"][k], out["precise"][k]
        assert a.shape == b.shape and a.dtype == b.dtype == torch.float16
        e = rel_l2(a, b)
        assert 0.0 < e < 2.5e-3, (k, e)              # different arithmetic (not bit-equal), same function


@pytest.mark.gpu
@pytest.mark.parametrize("base,lat,batch", [("xl", 16, 2), ("1-5", 16, 1)])
def test_split_operand_classes_each_and_selective(base, lat, batch):
    """Round 4: the split is per operand CLASS (include/gdf.h reserved[1] = mask << 8; components/native.py SPLIT_CLASSES).  Every single class
    builds and runs (a plan mixes plain and split tensors: each producer / consumer pair must agree on the row layout), stays within the
    default plan's tolerance, the SELECTIVE preset is measurably closer to the oracle than the default plan, and mask 255 == precise=True."""
    from components.native import SPLIT_CLASSES, SPLIT_SELECTIVE, SPLIT_ALL, split_mask
    arch = R.tiny_arch(base)
    P = R.synth_params(arch, seed=0)
    I = R.synth_inputs(arch, batch, lat, seed=1)
    ref = oracle_run(arch, P, I)
    ids = [k for k in ref.keys() if not k.endswith("-map")]
    _, dflt = run_native(native(arch, P, precise=False), I, ids)
    e_d = {k: rel_l2(dflt[k], ref[k]) for k in ids}
    med = lambda e: sorted(e.values())[len(e) // 2]
    for name, bit in SPLIT_CLASSES.items():
        u = native(arch, P, precise=name)
        assert u.split == bit and not u.precise
        _, h = run_native(u, I, ids)
        assert list(h.keys()) == ids
        e = {k: rel_l2(h[k], ref[k]) for k in ids}
        bad = {k: v for k, v in e.items() if not v < TOL}
        assert not bad, (name, sorted(bad.items(), key=lambda kv: -kv[1])[:5])
        assert med(e) < 1.1 * med(e_d), (name, med(e), med(e_d))
        print(f"[{base}] split class {name:9s}: median {med(e):.2e} worst {max(e.values()):.2e}   (default {med(e_d):.2e} / {max(e_d.values()):.2e})")
    us = native(arch, P, precise="selective")
    assert us.split == SPLIT_SELECTIVE == split_mask("selective")
    _, hs = run_native(us, I, ids)
    e_s = {k: rel_l2(hs[k], ref[k]) for k in ids}
    print(f"[{base}] selective: median {med(e_s):.2e} worst {max(e_s.values()):.2e}")
    assert med(e_s) < 0.85 * med(e_d) and max(e_s.values()) < max(e_d.values())
    ua, up = native(arch, P, precise=SPLIT_ALL), native(arch, P, precise=True)
    assert ua.precise and up.precise
    _, ha = run_native(ua, I, ids)
    _, hp = run_native(up, I, ids)
    for k in ids:
        assert torch.equal(ha[k], hp[k]), k


def test_hook_buffers_recycled_after_views_die_and_cross_stream_reader_is_ordered():
    """Hook-buffer lifetime on public API only: the returned tensors are views of a per-hand-out lease tensor; the set
    is recycled when the caller dropped every view (steady state: one set, graph replay), never while a view is alive, and a consumer on
    ANOTHER stream that announces its reads (components.native.release_after, the record_stream() of these buffers) may drop its views at
    once: the next forward — which overwrites the same buffers — is ordered behind the announced reads."""
    from components.native import release_after
    arch = R.tiny_arch("xl")
    P = R.synth_params(arch, seed=0)
    Ia, Ib = R.synth_inputs(arch, 2, 16, seed=1), R.synth_inputs(arch, 2, 16, seed=2)
    ids = ["mid-vit-block0-out", "up-level1-repeat0-vit-block0-cross-q"]
    u = native(arch, P)
    _, ha = run_native(u, Ia, ids)
    want_a = {k: v.clone() for k, v in ha.items()}
    _, hb = run_native(u, Ib, ids)                      # ha still alive: a second set
    plan = next(iter(u._plans.values()))
    assert len(plan.sets) == 2 and ha[ids[0]].data_ptr() != hb[ids[0]].data_ptr()
    for k in ids:
        assert torch.equal(ha[k], want_a[k])           # untouched by the second forward
    want_b = {k: v.clone() for k, v in hb.items()}
    view_of_view = ha[ids[0]][0].permute(1, 2, 0)       # a derived view keeps the set leased as well
    pa = ha[ids[0]].data_ptr()
    del ha
    _, hc = run_native(u, Ia, ids)
    assert len(plan.sets) == 3 and hc[ids[0]].data_ptr() != pa
    assert torch.equal(view_of_view, want_a[ids[0]][0].permute(1, 2, 0))
    del view_of_view, hc, hb
    # ---- a reader on a side stream, views dropped before its copy has run ----
    s2 = torch.cuda.Stream()
    host = {k: torch.empty(want_a[k].shape, dtype=torch.float16).pin_memory() for k in ids}
    _, ha = run_native(u, Ia, ids)
    s2.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s2):
        torch.cuda._sleep(400_000_000)                  # ~0.2 s: the copies below run long after the next forward was queued
        for k in ids:
            host[k].copy_(ha[k], non_blocking=True)
    release_after(ha, s2)
    p0 = ha[ids[0]].data_ptr()
    del ha, _                                           # (`_` held the model output of that forward: a view of the same lease)
    g = lambda k: Ib[k].cuda() if k in Ib else None
    _, hb2 = u.forward_raw(g("sample"), g("timestep"), g("ctx"), g("text_embeds"), g("time_ids"), hook_ids=ids)   # no synchronize
    assert hb2[ids[0]].data_ptr() == p0                 # the SAME buffers: recycled at once
    s2.synchronize(); torch.cuda.synchronize()
    for k in ids:
        assert torch.equal(host[k], want_a[k].cpu()), k     # the side-stream reader saw forward A's data
        assert torch.equal(hb2[k], want_b[k]), k


# ---- heavy-tailed weight statistics + the runtime self-check of the automatic plan level ----------------------
def _kind_table(errs):
    from oracle.operand_floor import kind_of
    kinds = {}
    for k, e in errs.items():
        kinds.setdefault(kind_of(k), []).append(e)
    return {kd: max(v) for kd, v in sorted(kinds.items())}


@pytest.mark.parametrize("base", ["xl", "1-5"])
def test_heavy_tailed_weights_plan_levels(base):
    """Synthetic weights with log-normal per-channel scales and a few x16 outlier channels in the residual stream
    (oracle/unet_ref.py synth_params_heavy — what real SD checkpoints look like and N(0, 1/fan_in) does not): the fp16-OPERAND FLOOR
    of such a model is several times that of the benign one, so the plain plan must not be trusted on table values alone.
    Reported: plain / selective / full-split error per hook kind; asserted: HIP plain plan stays within 2x of the fp16-operand
    floor of the SAME weights (it is not the kernels that lose the accuracy), every level improves on the one below, and the full split
    is clearly more accurate than the plain plan (what the split cannot remove is the fp16 storage of q / k / v / P inside attention,
    which peaky softmax rows amplify)."""
    from components.native import SELECTIVE_BY_ARCH, SPLIT_SELECTIVE, arch_family
    arch = R.tiny_arch(base)
    P = R.synth_params_heavy(arch, seed=0, outlier_gain=16.0)
    I = R.synth_inputs(arch, 1, 16, seed=1)
    ref = oracle_run(arch, P, I)
    ids = [k for k in ref if not k.endswith("-map")]
    assert all(torch.isfinite(v).all() for v in ref.values())
    assert max(float(v.abs().max()) for v in ref.values()) < 30000       # still inside the fp16 range, like a real checkpoint
    with fp16_operands():
        flo = oracle_run(arch, P, I)
    floor = {k: rel_l2(flo[k], ref[k]) for k in ids}
    sel = SELECTIVE_BY_ARCH.get(arch_family(cfg_from_oracle_arch(arch)), SPLIT_SELECTIVE)
    res = {}
    for name, spec in (("plain", False), ("selective", sel), ("precise", True)):
        _, hooks = run_native(native(arch, P, precise=spec), I, ids)
        res[name] = {k: rel_l2(hooks[k], ref[k]) for k in ids}
        print(f"[{base} heavy-tailed] {name:9s} worst {max(res[name].values()):.2e} median {sorted(res[name].values())[len(ids) // 2]:.2e}  "
              + "  ".join(f"{kd}={e:.1e}" for kd, e in _kind_table(res[name]).items()))
    print(f"[{base} heavy-tailed] fp16-operand floor worst {max(floor.values()):.2e} median {sorted(floor.values())[len(ids) // 2]:.2e}")
    # (with peaky softmax rows the flash kernel's fp16 P and the emulation's fp16 P round different numbers: within 2x, not within 1.3x as on benign weights)
    over = {k: (res["plain"][k], floor[k]) for k in ids if not res["plain"][k] <= 2.0 * floor[k] + 1e-4}
    assert not over, sorted(over.items(), key=lambda kv: -kv[1][0])[:8]
    med = lambda d: sorted(d.values())[len(d) // 2]
    assert med(res["precise"]) < med(res["selective"]) < med(res["plain"])
    assert max(res["precise"].values()) < max(res["plain"].values()) / 1.5


def test_verify_escalates_plan_on_heavy_tailed_weights(monkeypatch):
    """NativeUNet(verify=True) / GDF_VERIFY=1: the first forward of a hook set runs the table-chosen level AND the full split, compares the
    requested hooks and escalates when they differ by more than 9.5e-4.  Benign synthetic weights: the table's choice (plain) stands, no
    warning.  Heavy-tailed weights: the same hook set is escalated, one RuntimeWarning, the result handed out is the escalated plan's, and
    later forwards go straight to that level without a second check."""
    import warnings
    arch = R.tiny_arch("xl")
    I = R.synth_inputs(arch, 2, 16, seed=1)
    ids = ["down-level1-repeat0-vit-block0-out", "mid-vit-block0-self-q", "up-level1-repeat0-vit-block0-out"]
    # benign weights: at these SHRUNKEN widths the plain plan already differs from the full split by > 1e-3 (test_all_hooks: floor 1.5e-3), so the
    # check is exercised with a bound that the benign model passes and the heavy-tailed one does not
    # (this shrunken architecture is not in the table: its kind rules would start at the selective preset; start from the plain plan, as the
    #  table does for most hooks of the true architectures.  Round 6: an escalation is OR-ed onto the table's choice, so the table itself is
    #  stubbed rather than a zero escalation planted)
    import components.plan_levels as PL
    monkeypatch.setattr(PL, "choose_split", lambda cfg, hook_ids, lat=None: 0)
    ub = native(arch, R.synth_params(arch, seed=0), precise="auto", verify=True)
    ub.verify_bound = 3e-3
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        _, hb = run_native(ub, I, ids)
    assert not [x for x in w if "gdf verify" in str(x.message)]
    assert ub.verify_log and ub.verify_log[0][2] == 0 and ub.last_split == 0 and tuple(ids) in ub._verified
    _, plain = run_native(native(arch, R.synth_params(arch, seed=0), precise=False), I, ids)
    for k in ids:
        assert torch.equal(hb[k], plain[k])                                  # the plain plan's own result was handed out
    # heavy-tailed weights, same bound
    Ph = R.synth_params_heavy(arch, seed=0, outlier_gain=16.0)
    uh = native(arch, Ph, precise="auto", verify=True)
    uh.verify_bound = 3e-3
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        _, hh = run_native(uh, I, ids)
        n_plans = len(uh._plans)
        _, hh2 = run_native(uh, I, ids)                                      # second forward: no re-check, no new plan, same level
    msgs = [x for x in w if "gdf verify" in str(x.message)]
    assert len(msgs) == 1, [str(x.message) for x in w]
    key, seen, kept = uh.verify_log[0]
    assert len(uh.verify_log) == 1 and kept != 0 and seen[0] > 3e-3 and uh.last_split == kept and len(uh._plans) == n_plans
    # ending at the full split is reported for what it means: the check bounds the distance to the full split, not the full split's own error
    from components.native import SPLIT_ALL
    assert ("outside the statistics" in str(msgs[0].message)) == (kept == SPLIT_ALL)
    _, want = run_native(native(arch, Ph, precise=kept), I, ids)
    ref = oracle_run(arch, Ph, I, ids)
    for k in ids:
        assert torch.equal(hh[k], want[k]) and torch.equal(hh2[k], want[k])
    _, plain_h = run_native(native(arch, Ph, precise=False), I, ids)
    e_plain = max(rel_l2(plain_h[k], ref[k]) for k in ids)
    e_kept = max(rel_l2(hh[k], ref[k]) for k in ids)
    print(f"[verify] heavy-tailed tiny xl: plain plan {e_plain:.2e} -> kept mask {kept}: {e_kept:.2e} vs the fp32 oracle; differences to the full split {seen}")
    assert e_kept < e_plain


@pytest.mark.parametrize("base,n_ctx,same_prompt,precise", [("1-5", 154, True, False), ("xl", 91, False, False), ("xl", 231, True, True), ("2-1", 8, False, False),
                                                            ("1-5", 1, True, False)])
def test_prompt_embeddings_of_other_lengths_than_77(base, n_ctx, same_prompt, precise):
    """The reference hands LONGER prompt embeddings to the UNet when the prompt has more than 70 words (feature/diffusion_feature.py:165-171 ->
    components/encode_long_prompt.py:5-40: the token ids are encoded in chunks of 77 and concatenated, the last chunk as long as it happens to be), and
    callers may pass any `encoder_hidden_states` they like: the text length is a plan parameter here.  All hooks incl. the `cross-map`s (whose channel
    count IS the text length) against the oracle at 154 (two chunks), 91 / 231 (ragged last chunk; not a multiple of the 16-wide MFMA K step or of the
    32-key attention tile), 8 and 1 tokens; one prompt repeated (text K/V computed once per call) and per-sample prompts; default and full-split plans."""
    arch = R.tiny_arch(base)
    P = R.synth_params(arch, seed=0)
    I = R.synth_inputs(arch, 2, 16, seed=3, n_ctx=n_ctx, same_prompt=same_prompt)
    ref = oracle_run(arch, P, I)
    ids = list(ref.keys())
    u = native(arch, P, precise=precise)
    g = lambda k: I[k].cuda() if k in I else None
    noise, hooks = u.forward_raw(g("sample"), g("timestep"), g("ctx"), g("text_embeds"), g("time_ids"), hook_ids=ids, shared_ctx=same_prompt)
    torch.cuda.synchronize()
    assert list(hooks.keys()) == ids
    errs = {k: rel_l2(hooks[k], ref[k]) for k in ids}
    maps = [k for k in ids if k.endswith("cross-map")]
    assert maps and all(hooks[k].shape[-1] == n_ctx for k in maps)                     # (B, heads, queries, n_ctx): components/attention.py:238-244
    assert all(tuple(hooks[k].shape) == tuple(ref[k].shape) for k in ids)
    bad = {k: v for k, v in errs.items() if not v < (6e-4 if precise else TOL)}
    assert not bad, sorted(bad.items(), key=lambda kv: -kv[1])[:8]
    # every map row is a distribution over the n_ctx keys (padding keys of the last tile must not leak in)
    for k in maps[:3]:
        s = hooks[k].float().sum(-1)
        assert torch.allclose(s, torch.ones_like(s), atol=5e-3), k
