"""The branch a user with REAL checkpoints takes (-m gpu): `FeatureExtractor(layer, version, 'cuda')` WITHOUT
GDF_SYNTHETIC_WEIGHTS, i.e. components/models.py get_diffusion_model -> `diffusers.<Pipeline>.from_pretrained` ->
config_from_diffusers / flux_ / pixart_config_from_diffusers -> load_state_dict(pipe.unet.state_dict()) -> native VAE from
pipe.vae.state_dict() -> the scheduler objects' own `set_timesteps / get_timesteps / add_noise / scale_model_input / step`, with
`verify` at its real-checkpoint default (ON).  `diffusers` is tests/fake_diffusers (test infrastructure): stock call surface, module
shells with the reference's own parameter names / shapes / configs at the TRUE architectures, restated schedulers.

Reference being replaced: feature/components/models.py:18-56,150-172, feature/diffusion_feature.py:46-55 (constructor), :288-295 (timestep
selection), :371-380 (prepare_latents), :405-406 (scale_model_input), :446-474 (denoiser call), :477-485 (vae-out).

Every hook is compared with the fp32 CPU oracle on the SAME weights (read back from the fake pipeline's original modules), the same noisy
latents and the same conditioning; tolerance 1e-3 relative L2 (north star), stated per assertion.
"""
import json
import os
import subprocess
import sys
import warnings

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FAKE = os.path.join(ROOT, "tests", "fake_diffusers")
pytestmark = pytest.mark.gpu


@pytest.fixture()
def D(monkeypatch):
    """`import diffusers` -> tests/fake_diffusers, synthetic pipes OFF"""
    monkeypatch.syspath_prepend(FAKE)
    monkeypatch.delenv("GDF_SYNTHETIC_WEIGHTS", raising=False)
    monkeypatch.delenv("GDF_VERIFY", raising=False)
    monkeypatch.delenv("GDF_FLUX_DTYPE", raising=False)
    sys.modules.pop("diffusers", None)
    import diffusers
    assert diffusers.__version__.endswith("+fake")
    diffusers.reset()
    torch.set_num_threads(max(1, min(32, os.cpu_count() or 1)))
    yield diffusers
    diffusers.reset()
    sys.modules.pop("diffusers", None)
    torch.cuda.empty_cache()


def _images(n, size, seed=0):
    from PIL import Image
    rs = np.random.RandomState(seed)
    out = []
    for _ in range(n):
        # smooth random pictures (a few low-frequency waves + noise): every pixel value occurs, nothing saturates
        yy, xx = np.mgrid[0:size, 0:size].astype(np.float32) / size
        img = np.stack([0.5 + 0.35 * np.sin(6.3 * (rs.rand() * 3 * xx + rs.rand() * 3 * yy + rs.rand())) for _ in range(3)], -1)
        img = np.clip(img + 0.1 * rs.randn(size, size, 3), 0, 1)
        out.append(Image.fromarray((img * 255).astype(np.uint8)))
    return out


def _sd(module):
    return {k: v.detach().float().cpu() for k, v in module.state_dict().items()}


def _rel(got, ref):
    r = ref.to(got.device).float()
    return float((got.float() - r).norm() / (r.norm() + 1e-30))


def _record_prepare_latents(df):
    """wrap pipe.prepare_latents: keep its arguments, its result and the (eps, noise) the call drew from the CUDA generator"""
    rec = {}
    orig = df.pipe.prepare_latents

    def wrapped(image, timestep, batch_size, num_images_per_prompt, dtype, device, generator=None):
        state = torch.cuda.get_rng_state(device)
        out = orig(image, timestep, batch_size, num_images_per_prompt, dtype, device, generator)
        after = torch.cuda.get_rng_state(device)
        torch.cuda.set_rng_state(state, device)
        rec["eps"] = torch.randn(out.shape, device=device, dtype=torch.float32)
        rec["noise"] = torch.randn(out.shape, device=device, dtype=torch.float32)
        torch.cuda.set_rng_state(after, device)
        rec.update(image=image.detach().float().cpu(), timestep=timestep.clone(), latents=out.clone(), dtype=dtype)
        return out
    df.pipe.prepare_latents = wrapped
    return rec


def _unet_case(D, version, img, layer, B=2, t=100, verify=None, **ctor):
    import diffusion_feature
    from components.native import NativeUNet, NativeVAEEncoder
    with warnings.catch_warnings(record=True) as wlog:
        warnings.simplefilter("always")
        df = diffusion_feature.FeatureExtractor(layer=layer, version=version, device="cuda:0", img_size=img, verify=verify, **ctor)
        pipe = df.pipe
        assert D.PIPES[-1] is pipe and isinstance(pipe.unet, NativeUNet) and isinstance(pipe.native_vae, NativeVAEEncoder)
        assert not getattr(pipe, "synthetic_weights", False)
        rec = _record_prepare_latents(df)
        prompt = df.encode_prompt("a photo of a tabby cat on a sofa")
        imgs = _images(B, img, seed=3)
        feats = df.extract(prompt, batch_size=B, image=imgs, t=t)
        torch.cuda.synchronize()
        feats = {k: v.clone() for k, v in feats.items()}
    return df, pipe, rec, prompt, feats, wlog


def _check_vae_stage(D, pipe, rec, a, b, tol=1e-3):
    """noisy latents of the native VAE encoder (weights loaded from the fake AutoencoderKL's state_dict through load_vae_state_dict) vs the oracle"""
    from oracle import vae_ref as VR
    Pv = _sd(pipe.original["vae"])
    with torch.no_grad():
        want = VR.prepare_latents(Pv, VR.ARCH_SD_VAE, rec["image"], rec["eps"].cpu(), rec["noise"].cpu(), float(pipe.vae.config.scaling_factor), a, b)
    e = _rel(rec["latents"], want)
    assert e <= tol, e
    return e


def test_sdxl_real_checkpoint_branch(D):
    """'xl': StableDiffusionXLImg2ImgPipeline.from_pretrained(variant='fp16') with the checkpoint's OWN scheduler (Euler for sd_xl_base) -> native UNet + VAE;
    512^2, B = 2, a layer set that selects the selective split and therefore a verify run against the full split."""
    from oracle import unet_ref as R
    import conftest
    layer = {"down-level1-repeat0-vit-block0-out": True, "mid-vit-block3-ffn-inner": True, "up-level0-repeat1-vit-block5-self-q": True,
             "up-level0-repeat2-vit-block9-cross-q": True, "up-level1-repeat2-res-out": True, "up-level2-repeat1-res-increment": True,
             "up-level1-repeat0-vit-out": True, "unet-out": True}
    df, pipe, rec, prompt, feats, wlog = _unet_case(D, "xl", 512, layer)
    calls = dict(D.CALLS)
    fp = calls["StableDiffusionXLImg2ImgPipeline.from_pretrained"]
    assert fp["repo"] == "stabilityai/stable-diffusion-xl-base-1.0" and fp["variant"] == "fp16" and fp["torch_dtype"] == torch.float16 and fp["use_safetensors"]
    assert "unet" not in fp                                              # single process: every component is loaded
    # reference models.py:43-56: no scheduler swap for 'xl' — the pipeline keeps the one its checkpoint names (sd_xl_base: EulerDiscreteScheduler)
    assert not any(c.startswith("EulerDiscreteScheduler.") for c in calls) and isinstance(pipe.scheduler, D.EulerDiscreteScheduler)
    u = pipe.unet
    from components.native import ARCH_CONFIGS, SELECTIVE_BY_ARCH
    assert {k: (tuple(v) if isinstance(v, (list, tuple)) else v) for k, v in u.cfg.items()} == \
        {k: (tuple(v) if isinstance(v, (list, tuple)) else v) for k, v in ARCH_CONFIGS["xl"].items()}
    assert u.verify is True                                              # the real-checkpoint default (diffusion_feature.py verify=None)
    assert list(feats.keys()) == [k for k in open(os.path.join(ROOT, "tests/golden/ids_xl_full.txt")).read().split() if k in layer]   # execution order
    # ---- timestep / scheduler scalars, as the scheduler object itself defines them ----
    t = rec["timestep"]
    assert t.shape == (2,) and float(t[0]) == 100.0
    sigma = float(pipe.scheduler.sigmas[pipe.scheduler.index_for_timestep(t[0])])
    # ---- stage 1: VAE encode + sample + add_noise ----
    e_vae = _check_vae_stage(D, pipe, rec, 1.0, sigma)
    # ---- stage 2: UNet on the native stage-1 latents ----
    arch = R.ARCHS["xl"]
    P = _sd(pipe.original["unet"])
    lat_in = (rec["latents"].float() / (sigma ** 2 + 1) ** 0.5).cpu()
    ids = list(feats.keys())
    st = R.Store({k: True for k in ids})
    pe, _, pooled, _ = prompt
    time_ids = torch.tensor([[512.0, 512.0, 0.0, 0.0, 512.0, 512.0]]).repeat(2, 1)
    with torch.no_grad():
        R.unet_forward(P, arch, lat_in, torch.tensor([100.0]), pe.float().cpu().repeat(2, 1, 1), pooled.float().cpu().repeat(2, 1), time_ids, store=st)
    errs = {k: _rel(feats[k], st.feats[k]) for k in ids}
    print(f"\n[fake-diffusers xl 512^2 B=2] VAE stage {e_vae:.2e}; hooks: " + ", ".join(f"{k.split('-', 2)[-1]} {e:.2e}" for k, e in errs.items()))
    assert max(errs.values()) <= 1.0e-3, errs                            # north star, every requested hook (the plan chooser's contract)
    for k in ids:
        assert feats[k].dtype == torch.float16 and feats[k].shape == st.feats[k].shape
    # ---- verify ran ONCE for this layer set, against the full split, and kept a level; its check-only plans are gone ----
    assert len(u.verify_log) == 1 and [set(k) for k in u._verified] == [set(ids)]
    key, seen, kept = u.verify_log[0]
    assert kept in (SELECTIVE_BY_ARCH["xl"], u.last_split) and seen[kept] <= u.verify_accept_bound()
    assert abs(u.verify_accept_bound() - (0.97e-3 ** 2 - 2.7e-4 ** 2) ** 0.5) < 1e-9
    assert {k[8] for k in u._plans} == {kept}
    conftest.record_margin("fake-diffusers xl 512^2 B=2 (real-checkpoint branch, verify ON)", max(errs, key=errs.get), max(errs.values()), 1.0e-3,
                           extra=f"VAE stage {e_vae:.2e}; verify d={seen[kept]:.2e} kept {kept}")
    # a second extract of the same layer set: no second verification, same bits
    n_log = len(u.verify_log)
    torch.manual_seed(1)
    f2 = df.extract(prompt, batch_size=2, image=rec["latents"], image_type="latents", t=100)
    f3 = df.extract(prompt, batch_size=2, image=rec["latents"], image_type="latents", t=100)
    torch.cuda.synchronize()
    assert len(u.verify_log) == n_log and all(torch.equal(f2[k], f3[k]) for k in ids)
    assert all(torch.equal(f2[k], feats[k]) for k in ids)                # image -> latents -> hooks == latents -> hooks


def test_sdxl_real_checkpoint_headline_shape_verify_completes(D):
    """The headline shape on the real-checkpoint branch: SDXL 1024^2, B = 16, the four `config_xl_practical` hooks, `verify` at its default (ON).
    The first batch builds TWO plans at B = 16 (the plain plan the table picks and the full split it is checked against: the full split must exist
    at this size — 32-bit buffer offsets — and fit), the check completes without a warning and keeps the plain plan, the check-only plan is released
    and the second batch replays one graph."""
    import diffusion_feature
    layer = {k: True for k in ("up-level0-repeat0-vit-block7-out", "up-level0-repeat0-vit-block5-out", "up-level1-repeat0-vit-block0-cross-q",
                               "up-level1-repeat0-vit-block0-out")}
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        df = diffusion_feature.FeatureExtractor(layer=layer, version="xl", device="cuda:0", img_size=1024)
        u = df.pipe.unet
        assert u.verify is True
        prompt = df.encode_prompt("a photo of a red fox in the snow")
        x = torch.rand(16, 3, 1024, 1024, generator=torch.Generator().manual_seed(0)) * 2 - 1
        f1 = {k: v.clone() for k, v in df.extract(prompt, batch_size=16, image=x, image_type="tensors", t=100).items()}
        torch.cuda.synchronize()
    assert not [m for m in w if "gdf verify" in str(m.message)], [str(m.message) for m in w]
    assert len(u.verify_log) == 1 and u.verify_log[0][2] == 0 and u.last_split == 0          # the plain plan stands: d <= sqrt((0.97e-3)^2 - (2.7e-4)^2)
    d = u.verify_log[0][1][0]
    assert d <= u.verify_accept_bound()
    assert {k[8] for k in u._plans} == {0} and all(k[0] == 16 for k in u._plans)
    plan = next(iter(u._plans.values()))
    cap0 = plan.graph_stats()
    torch.manual_seed(1)
    f2 = df.extract(prompt, batch_size=16, image=x, image_type="tensors", t=100)
    torch.cuda.synchronize()
    cap1 = plan.graph_stats()
    assert len(u.verify_log) == 1 and cap1[2] == 0 and cap1[1] > cap0[1]                     # no second check; graph replays, no eager fallback
    for k in layer:
        assert f2[k].shape == f1[k].shape and f2[k].shape[0] == 16 and torch.isfinite(f2[k].float()).all()
    import conftest
    conftest.record_margin("fake-diffusers xl 1024^2 B=16 practical hooks: verify's distance plain -> full split", "worst of 4 hooks", d, u.verify_accept_bound())


def test_sd15_real_checkpoint_branch_with_vae_out(D):
    """'1-5': PNDM (alphas_cumprod add_noise, identity scale_model_input, step_plms probed for 'vae-out'); int-valued config fields
    (attention_head_dim = 8, transformer_layers_per_block = 1); 256^2, B = 2."""
    from oracle import unet_ref as R, vae_ref as VR
    import conftest
    layer = {"down-level0-repeat1-vit-block0-self-k": True, "mid-vit-block0-ffn-inner": True, "up-level1-repeat1-vit-block0-cross-q": True,
             "up-level2-repeat2-res-out": True, "up-level3-repeat2-vit-block0-out": True, "unet-out": True, "vae-out": True}
    df, pipe, rec, prompt, feats, wlog = _unet_case(D, "1-5", 256, layer)
    calls = dict(D.CALLS)
    fp = calls["StableDiffusionImg2ImgPipeline.from_pretrained"]
    assert fp["repo"] == "stable-diffusion-v1-5/stable-diffusion-v1-5" and "variant" not in fp
    assert "EulerDiscreteScheduler.from_config" not in calls and isinstance(pipe.scheduler, D.PNDMScheduler)
    assert not pipe.unet.weights_rounded and not any("rounded once at load" in str(w.message) for w in wlog)      # an fp16 checkpoint is stored exactly
    assert pipe.unet.cfg["heads"] == (8, 8, 8, 8) and pipe.unet.cfg["transformer_layers"] == (1, 1, 1, 1) and not pipe.unet.cfg["use_linear_projection"]
    t = rec["timestep"]
    assert int(t[0]) == 101                                              # PNDM's leading spacing + steps_offset 1
    ac = pipe.scheduler.alphas_cumprod.double()
    a, b = float(ac[101] ** 0.5), float((1 - ac[101]) ** 0.5)
    e_vae = _check_vae_stage(D, pipe, rec, a, b)
    arch = R.ARCHS["1-5"]
    P = _sd(pipe.original["unet"])
    ids = [k for k in feats if k != "vae-out"]
    st = R.Store({k: True for k in ids})
    pe = prompt[0]
    with torch.no_grad():
        R.unet_forward(P, arch, rec["latents"].float().cpu(), torch.tensor([101.0]), pe.float().cpu().repeat(2, 1, 1), store=st)
    errs = {k: _rel(feats[k], st.feats[k]) for k in ids}
    assert max(errs.values()) <= 1.0e-3, errs
    # ---- 'vae-out' (reference :477-485): scheduler.step(noise_pred, t, latents)[0] by the scheduler OBJECT, then decode / scaling_factor ----
    sch = D.PNDMScheduler(**{k: v for k, v in pipe.scheduler.config.items() if not k.startswith("_")})
    sch.set_timesteps(1000, device="cpu")
    noise_pred = feats["unet-out"].float().cpu()                         # the native noise prediction (stage separation: its own error is asserted above)
    with torch.no_grad():
        prev = sch.step(noise_pred.double(), t[:1].cpu(), rec["latents"].double().cpu(), return_dict=False)[0].float()
        want = VR.decode(_sd(pipe.original["vae"]), VR.ARCH_SD_VAE, prev / float(pipe.vae.config.scaling_factor))
    e_out = _rel(feats["vae-out"], want)
    assert tuple(feats["vae-out"].shape) == (2, 3, 256, 256) and e_out <= 1.0e-3, e_out
    print(f"\n[fake-diffusers 1-5 256^2 B=2] VAE stage {e_vae:.2e}; vae-out {e_out:.2e}; hooks: " + ", ".join(f"{k.split('-', 2)[-1]} {e:.2e}" for k, e in errs.items()))
    conftest.record_margin("fake-diffusers 1-5 256^2 B=2 + vae-out (PNDM probes)", max(errs, key=errs.get), max(errs.values()), 1.0e-3,
                           extra=f"VAE stage {e_vae:.2e}, vae-out {e_out:.2e}")


def test_sd21_real_checkpoint_branch(D):
    """'2-1': Euler made from the PNDM-family scheduler config of the checkpoint (`EulerDiscreteScheduler.from_pretrained(model_id, subfolder="scheduler")`
    handed to the pipeline's from_pretrained, reference models.py:38-39),
    64-wide heads (5, 10, 20, 20), linear proj_in / proj_out, OpenCLIP width 1024."""
    from oracle import unet_ref as R
    layer = {"down-level1-repeat1-vit-block0-out": True, "up-level1-repeat0-vit-block0-self-v": True, "up-level3-repeat1-res-out": True}
    df, pipe, rec, prompt, feats, wlog = _unet_case(D, "2-1", 256, layer, verify=False)
    calls = dict(D.CALLS)
    assert calls["StableDiffusionImg2ImgPipeline.from_pretrained"]["repo"] == "stabilityai/stable-diffusion-2-1-base"
    assert calls["EulerDiscreteScheduler.from_pretrained"] == dict(repo="stabilityai/stable-diffusion-2-1-base", subfolder="scheduler")
    assert isinstance(calls["StableDiffusionImg2ImgPipeline.from_pretrained"]["scheduler"], D.EulerDiscreteScheduler) and isinstance(pipe.scheduler, D.EulerDiscreteScheduler)
    assert pipe.scheduler.config.steps_offset == 1 and pipe.scheduler.config.timestep_spacing == "leading"
    assert pipe.unet.cfg["heads"] == (5, 10, 20, 20) and pipe.unet.cfg["cross_attention_dim"] == 1024 and pipe.unet.cfg["use_linear_projection"] == 1
    t = rec["timestep"]
    sigma = float(pipe.scheduler.sigmas[pipe.scheduler.index_for_timestep(t[0])])
    _check_vae_stage(D, pipe, rec, 1.0, sigma)
    st = R.Store({k: True for k in feats})
    with torch.no_grad():
        R.unet_forward(_sd(pipe.original["unet"]), R.ARCHS["2-1"], (rec["latents"].float() / (sigma ** 2 + 1) ** 0.5).cpu(), torch.tensor([float(t[0])]),
                       prompt[0].float().cpu().repeat(2, 1, 1), store=st)
    errs = {k: _rel(feats[k], st.feats[k]) for k in feats}
    assert max(errs.values()) <= 1.0e-3, errs


def test_float32_dtype_real_checkpoint_branch(D):
    """dtype='float32' (reference feature/components/models.py:11-12: the pipeline in torch.float32; the FeatureStore still hands out fp16, feature_extractor.py
    :59-60): the front-end modules are fp32 (prompt embeddings, the latents handed to the UNet), the fp32 state dict reaches libgdf, the UNet runs the
    FULL-SPLIT operand plan, and an explicit `precise=` still wins.  What the mode can and cannot promise is asserted both ways: the native UNet keeps ONE
    fp16 image per weight matrix, so it computes the fp16-ROUNDED checkpoint (the weights the reference's dtype='float16' mode uses) — within the full
    split's own bound (6e-4, SD1.5 family) of the oracle on those weights — and is therefore ~1e-3 from the oracle on the fp32 weights (bound 1.35e-3 stated
    here, one warning at load)."""
    import diffusion_feature
    from components import plan_levels as PL
    from oracle import unet_ref as R
    import conftest
    layer = {"down-level1-repeat1-vit-block0-ffn-inner": True, "mid-vit-block0-out": True, "up-level2-repeat1-vit-block0-self-q": True,
             "up-level3-repeat2-res-out": True, "unet-out": True}
    df, pipe, rec, prompt, feats, wlog = _unet_case(D, "1-5", 256, layer, dtype="float32")
    fp = dict(D.CALLS)["StableDiffusionImg2ImgPipeline.from_pretrained"]
    assert fp["torch_dtype"] == torch.float32
    assert next(pipe.original["unet"].parameters()).dtype == torch.float32 and prompt[0].dtype == torch.float32 and rec["dtype"] == torch.float32
    assert not pipe.unet.auto_split and pipe.unet.split == PL.SPLIT_ALL and pipe.unet.last_split == PL.SPLIT_ALL
    assert all(v.dtype == torch.float16 for v in feats.values())
    said = [str(w.message) for w in wlog if "rounded once at load" in str(w.message)]
    assert pipe.unet.weights_rounded and sum("NativeUNet" in m for m in said) == 1 and sum("NativeVAEEncoder" in m for m in said) == 1
    ac = pipe.scheduler.alphas_cumprod.double()
    _check_vae_stage(D, pipe, rec, float(ac[101] ** 0.5), float((1 - ac[101]) ** 0.5))
    P32 = _sd(pipe.original["unet"])
    P16 = {k: (v.half().float() if v.dim() >= 2 else v) for k, v in P32.items()}      # matrices rounded once, vectors kept (the arena keeps those fp32)
    errs = {}
    for tag, P in (("fp16-rounded weights", P16), ("fp32 weights", P32)):
        st = R.Store({k: True for k in feats})
        with torch.no_grad():
            R.unet_forward(P, R.ARCHS["1-5"], rec["latents"].float().cpu(), torch.tensor([101.0]), prompt[0].float().cpu().repeat(2, 1, 1), store=st)
        errs[tag] = {k: _rel(feats[k], st.feats[k]) for k in feats}
    w16, w32 = max(errs["fp16-rounded weights"].values()), max(errs["fp32 weights"].values())
    assert w16 <= 6.0e-4, errs
    assert w32 <= 1.35e-3, errs
    conftest.record_margin("fake-diffusers 1-5 256^2 B=2 dtype='float32' -> full split (vs the fp16-rounded checkpoint)",
                           max(errs["fp16-rounded weights"], key=errs["fp16-rounded weights"].get), w16, 6.0e-4, extra=f"vs the fp32 weights {w32:.2e} / 1.35e-3")
    del df, pipe, feats
    D.reset()
    df = diffusion_feature.FeatureExtractor(layer=layer, version="1-5", device="cuda:0", img_size=256, dtype="float32", precise="auto", verify=False)
    assert df.pipe.unet.auto_split


def test_float32_dtype_xl_loads_the_fp16_variant_and_is_exact_in_the_arena(D):
    """'xl' / 'pgv2' load `variant="fp16"` whatever dtype says (reference feature/components/models.py:51-53, 65-67): under dtype='float32' the fp32 modules
    hold fp16-exact numbers, the native arena stores them without loss (no warning), and the full split is within 7e-4 of the oracle on the SAME fp32
    weights and the SAME fp32 inputs — i.e. of what the reference's float32 mode computes.  (Measured 3.3-5.1e-4: the full split's own 1.9-2.7e-4 plus the
    one fp16 rounding of the fp32 latents / prompt embeddings / pooled embeddings at the C-ABI boundary, whose operand images are 16-bit.)"""
    from components import plan_levels as PL
    from oracle import unet_ref as R
    import conftest
    layer = {"down-level2-repeat1-vit-block3-ffn-inner": True, "mid-vit-block5-out": True, "up-level0-repeat2-vit-block9-self-q": True,
             "up-level1-repeat2-res-out": True, "unet-out": True}
    df, pipe, rec, prompt, feats, wlog = _unet_case(D, "xl", 512, layer, B=1, dtype="float32")
    fp = dict(D.CALLS)["StableDiffusionXLImg2ImgPipeline.from_pretrained"]
    assert fp["variant"] == "fp16" and fp["torch_dtype"] == torch.float32
    assert next(pipe.original["unet"].parameters()).dtype == torch.float32 and prompt[0].dtype == torch.float32
    assert not pipe.unet.weights_rounded and not any("rounded once at load" in str(w.message) for w in wlog)
    assert pipe.unet.last_split == PL.SPLIT_ALL and all(v.dtype == torch.float16 for v in feats.values())
    t = rec["timestep"]
    sigma = float(pipe.scheduler.sigmas[pipe.scheduler.index_for_timestep(t[0])])
    _check_vae_stage(D, pipe, rec, 1.0, sigma)
    st = R.Store({k: True for k in feats})
    tid = torch.tensor([[512.0, 512.0, 0.0, 0.0, 512.0, 512.0]])
    with torch.no_grad():
        R.unet_forward(_sd(pipe.original["unet"]), R.ARCHS["xl"], (rec["latents"].float() / (sigma ** 2 + 1) ** 0.5).cpu(), torch.tensor([float(t[0])]),
                       prompt[0].float().cpu(), prompt[2].float().cpu(), tid, store=st)
    errs = {k: _rel(feats[k], st.feats[k]) for k in feats}
    assert max(errs.values()) <= 7.0e-4, errs
    conftest.record_margin("fake-diffusers xl 512^2 B=1 dtype='float32' (fp16 variant upcast) -> full split", max(errs, key=errs.get), max(errs.values()), 7.0e-4)


def test_long_prompt_embeddings_reach_the_unet(D):
    """A prompt of more than 70 words (reference feature/diffusion_feature.py:165-171): encode_prompt returns (1, n_tokens, C) embeddings from the windowed
    text encoder and no pooled embeddings; extract() repeats them over the batch and the native UNet runs with n_tokens keys in every cross-attention —
    hooks incl. a `cross-map` (whose last axis IS the text length) against the oracle on the same embeddings."""
    import diffusion_feature
    from oracle import unet_ref as R
    layer = {"down-level1-repeat0-vit-block0-cross-map": True, "mid-vit-block0-cross-q": True, "up-level2-repeat1-vit-block0-out": True, "unet-out": True}
    df = diffusion_feature.FeatureExtractor(layer=layer, version="1-5", device="cuda:0", img_size=256, verify=False)
    prompt = df.encode_prompt(" ".join(f"word{i}" for i in range(120)))      # 122 ids = 77 + 45
    assert prompt[0].shape == (1, 122, 768) and prompt[0].dtype == torch.float16 and prompt[2] is None
    assert [c[1]["tokens"] for c in D.CALLS if c[0] == "text_encoder"] == [77, 45, 77, 45] and "encode_prompt" not in [c[0] for c in D.CALLS]
    lat = torch.randn(2, 4, 32, 32, generator=torch.Generator().manual_seed(5)).half()
    feats = df.extract(prompt, batch_size=2, image=lat, image_type="latents", t=100)
    torch.cuda.synchronize()
    assert feats["down-level1-repeat0-vit-block0-cross-map"].shape == (2, 8, 256, 122)
    st = R.Store({k: True for k in layer})
    with torch.no_grad():
        R.unet_forward(_sd(df.pipe.original["unet"]), R.ARCHS["1-5"], lat.float(), torch.tensor([101.0]), prompt[0].float().cpu().repeat(2, 1, 1), store=st)
    errs = {k: _rel(feats[k], st.feats[k]) for k in layer}
    assert max(errs.values()) <= 1.0e-3, errs


def test_offline_lora_is_fused_before_the_weights_are_read_and_native_vae_opt_out(D, monkeypatch):
    """components/models.py: `offline_lora` -> pipe.load_lora_weights(path, weight_name=...) + pipe.fuse_lora() BEFORE the UNet's state dict is handed
    to libgdf (reference feature/diffusion_feature.py:46-55 loads the LoRA into the pipeline), so the native model computes with the FUSED weights;
    and GDF_NATIVE_VAE=0 leaves the pipeline's own prepare_latents in place."""
    import diffusion_feature
    from oracle import unet_ref as R
    layer = {"up-level1-repeat1-vit-block0-out": True, "up-level2-repeat0-res-out": True}
    lat = torch.randn(2, 4, 32, 32, generator=torch.Generator().manual_seed(0)).half()
    outs = {}
    for tag, kw in (("plain", {}), ("lora", dict(offline_lora="/data/lora/dir", offline_lora_filename="pytorch_lora_weights.safetensors"))):
        D.reset()
        df = diffusion_feature.FeatureExtractor(layer=layer, version="1-5", device="cuda:0", img_size=256, verify=False, **kw)
        calls = [c[0] for c in D.CALLS]
        assert ("load_lora_weights" in calls and "fuse_lora" in calls) == (tag == "lora")
        if tag == "lora":
            assert dict(D.CALLS)["load_lora_weights"] == dict(path="/data/lora/dir", weight_name="pytorch_lora_weights.safetensors")
            assert calls.index("fuse_lora") > calls.index("load_lora_weights") > calls.index("StableDiffusionImg2ImgPipeline.from_pretrained")
        prompt = df.encode_prompt("a photo of a cat")
        f = df.extract(prompt, batch_size=2, image=lat, image_type="latents", t=100)
        torch.cuda.synchronize()
        outs[tag] = {k: v.clone() for k, v in f.items()}
        if tag == "lora":
            st = R.Store({k: True for k in layer})
            with torch.no_grad():                                            # the oracle on the weights of the FUSED module
                R.unet_forward(_sd(df.pipe.original["unet"]), R.ARCHS["1-5"], lat.float(), torch.tensor([101.0]), prompt[0].float().cpu().repeat(2, 1, 1), store=st)
            errs = {k: _rel(f[k], st.feats[k]) for k in layer}
            assert max(errs.values()) <= 1.0e-3, errs
        del df, f
    assert all(not torch.equal(outs["plain"][k], outs["lora"][k]) for k in layer)          # the fused weight did reach the kernels
    # ---- `offline_lora` WITHOUT a file name is a local MODEL directory that replaces the hub id, and no LoRA call is made (reference
    #      feature/components/models.py:21-22 and every other version branch; diffusion_feature.py:50-53 "else: TODO") ----
    D.reset()
    df = diffusion_feature.FeatureExtractor(layer=layer, version="1-5", device="cuda:0", img_size=256, verify=False, offline_lora="/data/models/my-finetuned-sd15")
    calls = [c[0] for c in D.CALLS]
    assert dict(D.CALLS)["StableDiffusionImg2ImgPipeline.from_pretrained"]["repo"] == "/data/models/my-finetuned-sd15"
    assert "load_lora_weights" not in calls and "fuse_lora" not in calls
    f = df.extract(df.encode_prompt("a photo of a cat"), batch_size=2, image=lat, image_type="latents", t=100)
    torch.cuda.synchronize()
    assert all(torch.equal(f[k], outs["plain"][k]) for k in layer)         # (the fake builds the same seeded weights whatever the directory is called)
    del df, f
    # ---- GDF_NATIVE_VAE=0: diffusers' own prepare_latents stays (the fake's raises if called; constructing the extractor must not touch it) ----
    monkeypatch.setenv("GDF_NATIVE_VAE", "0")
    D.reset()
    df = diffusion_feature.FeatureExtractor(layer=layer, version="1-5", device="cuda:0", img_size=256, verify=False)
    assert getattr(df.pipe, "native_vae", None) is None and df.pipe.prepare_latents.__func__ is D.StableDiffusionImg2ImgPipeline.prepare_latents
    with pytest.raises(AssertionError, match="torch VAE encode does not exist"):
        df.extract(df.encode_prompt("x"), batch_size=1, image=_images(1, 256), t=100)


def test_real_checkpoint_branch_accept_all_resize_attention_and_pgv2(D):
    """The remaining constructor surface on the real-checkpoint branch: `layer=None` (accept-all: every id of the architecture in execution order incl.
    the `*-map` hooks, handed out as CPU tensors, reference feature_extractor.py:9-15,65-66), `feature_resize=2` (:51-53), `attention=[...]` (the
    aggregated `attn` feature, diffusion_feature.py:492-500), and version 'pgv2' (Playground-v2: the SDXL pipeline class on another repo id, Euler
    the checkpoint's own scheduler; reference models.py:55-68)."""
    import diffusion_feature
    ids15 = open(os.path.join(ROOT, "tests/golden/ids_15_full.txt")).read().split()
    df = diffusion_feature.FeatureExtractor(layer=None, version="1-5", device="cuda:0", img_size=256, feature_resize=2, attention=["up_cross", "down_self"], verify=False)
    feats = df.extract(df.encode_prompt("a photo of a cat"), batch_size=1, image=_images(1, 256, seed=1), t=50)
    keys = list(feats.keys())
    assert keys[-1] == "attn" and keys[:-1] == ids15                         # == the key order of the reference's config_15_full.json
    assert all(v.device.type == "cpu" for k, v in feats.items() if k != "attn")
    assert feats["up-level2-repeat2-res-out"].shape == (1, 640, 8, 8)        # 16 x 16 at 256^2 (latent 32, one level down), pooled by feature_resize = 2
    assert feats["attn"].shape[0] == 1 and feats["attn"].shape[-2:] == (32, 32) and torch.isfinite(feats["attn"].float()).all()
    del df, feats
    D.reset()
    df = diffusion_feature.FeatureExtractor(layer={"up-level0-repeat1-vit-block2-out": True}, version="pgv2", device="cuda:0", img_size=256, verify=False)
    calls = dict(D.CALLS)
    assert calls["StableDiffusionXLImg2ImgPipeline.from_pretrained"]["repo"] == "playgroundai/playground-v2-1024px-aesthetic"
    assert calls["StableDiffusionXLImg2ImgPipeline.from_pretrained"]["variant"] == "fp16" and not any(c.startswith("EulerDiscreteScheduler.") for c in calls)
    f = df.extract(df.encode_prompt("x"), batch_size=2, image=_images(2, 256, seed=2), t=100)
    torch.cuda.synchronize()
    assert f["up-level0-repeat1-vit-block2-out"].shape == (2, 1280, 8, 8) and torch.isfinite(f["up-level0-repeat1-vit-block2-out"].float()).all()


def test_pixart_sigma_real_checkpoint_branch(D):
    """'pixart-sigma' from a STOCK (text-to-image) PixArtSigmaPipeline: the product supplies get_timesteps and the image-taking prepare_latents
    (native VAE + DPMSolverMultistep add_noise), builds the DiT from `pipe.transformer.config`, drops nothing but the `pos_embed.pos_embed` buffer."""
    import diffusion_feature
    from components.native import NativePixArtTransformer, PIXART_CONFIGS
    from oracle import pixart_ref as PR
    import conftest
    layer = {"vit-block0-self-q": True, "vit-block13-cross-q": True, "vit-block20-ffn-inner": True, "vit-block27-out": True}
    df = diffusion_feature.FeatureExtractor(layer=layer, version="pixart-sigma", device="cuda:0", img_size=512)
    pipe = df.pipe
    assert isinstance(pipe.transformer, NativePixArtTransformer) and pipe.unet is pipe.transformer and pipe.transformer.cfg == PIXART_CONFIGS["pixart-sigma"]
    assert dict(D.CALLS)["PixArtSigmaPipeline.from_pretrained"]["repo"] == "PixArt-alpha/PixArt-Sigma-XL-2-1024-MS"
    rec = _record_prepare_latents(df)
    prompt = df.encode_prompt("a watercolor painting of a lighthouse at dusk")
    assert len(prompt) == 4 and prompt[1].shape == (1, 300) and int(prompt[1].sum()) == 10
    feats = df.extract(prompt, batch_size=2, image=_images(2, 512, seed=5), t=261)
    torch.cuda.synchronize()
    t = rec["timestep"]
    acl = torch.cumprod(1 - torch.linspace(0.0001, 0.02, 1000, dtype=torch.float32), 0).double()
    tt = int(t[0])
    assert abs(tt - 261) <= 1
    e_vae = _check_vae_stage(D, pipe, rec, float(acl[tt] ** 0.5), float((1 - acl[tt]) ** 0.5))
    st = PR.Store({k: True for k in feats})
    P = _sd(pipe.original["transformer"])
    with torch.no_grad():
        PR.pixart_forward(P, PR.ARCH_PIXART_SIGMA, rec["latents"].float().cpu(), prompt[0].float().cpu().repeat(2, 1, 1), torch.tensor([float(tt)] * 2),
                          prompt[1].cpu().repeat(2, 1), st, want_map=False)
    errs = {k: _rel(feats[k], st.feats[k]) for k in feats}
    print(f"\n[fake-diffusers pixart-sigma 512^2 B=2] VAE stage {e_vae:.2e}; hooks: {errs}")
    assert list(feats) == list(st.feats) and max(errs.values()) <= 1.0e-3, errs
    conftest.record_margin("fake-diffusers pixart-sigma 512^2 B=2 (stock text-to-image pipeline)", max(errs, key=errs.get), max(errs.values()), 1.0e-3,
                           extra=f"VAE stage {e_vae:.2e}")


@pytest.mark.parametrize("version,repo,cls,n_txt", [("pixart-alpha", "PixArt-alpha/PixArt-XL-2-512x512", "PixArtAlphaPipeline", 120),
                                                   ("pixart-sigma-512", "PixArt-alpha/PixArt-Sigma-XL-2-512-MS", "PixArtSigmaPipeline", 300)])
def test_pixart_512_variants_real_checkpoint_branch(D, monkeypatch, version, repo, cls, n_txt):
    """the two 512-pixel PixArt versions of the reference (models.py:88-115): repo ids, pipeline classes, `variant='fp16'` for alpha only, the
    sample_size 64 / interpolation_scale 1 architecture read from the loaded module's config, caption lengths 120 / 300 (depth cut to 2 blocks)."""
    import diffusion_feature
    from oracle import pixart_ref as PR
    monkeypatch.setenv("FAKE_DIFFUSERS_PIXART_LAYERS", "2")
    layer = {"vit-block0-cross-q": True, "vit-block1-out": True}
    df = diffusion_feature.FeatureExtractor(layer=layer, version=version, device="cuda:0", img_size=256)
    fp = dict(D.CALLS)[cls + ".from_pretrained"]
    assert fp["repo"] == repo and (fp.get("variant") == "fp16") == (version == "pixart-alpha")
    tr = df.pipe.transformer
    assert tr.cfg["sample_size"] == 64 and tr.cfg["interpolation_scale"] == 1 and tr.cfg["num_layers"] == 2
    rec = _record_prepare_latents(df)
    prompt = df.encode_prompt("a red bicycle")
    assert prompt[0].shape == (1, n_txt, 4096)
    feats = df.extract(prompt, batch_size=2, image=_images(2, 256, seed=2), t=400)
    torch.cuda.synchronize()
    arch = dict(PR.ARCH_PIXART_SIGMA, num_layers=2, sample_size=64, interpolation_scale=1)
    st = PR.Store({k: True for k in feats})
    with torch.no_grad():
        PR.pixart_forward(_sd(df.pipe.original["transformer"]), arch, rec["latents"].float().cpu(), prompt[0].float().cpu().repeat(2, 1, 1),
                          torch.tensor([float(rec["timestep"][0])] * 2), prompt[1].cpu().repeat(2, 1), st, want_map=False)
    errs = {k: _rel(feats[k], st.feats[k]) for k in feats}
    assert max(errs.values()) <= 1.0e-3, errs


def test_flux_offline_lora_is_fused_before_the_weights_are_read(D, monkeypatch):
    """flux branch of get_diffusion_model with an offline LoRA (reference models.py:150-172 + diffusion_feature.py:46-55): loaded and fused into the
    pipeline's transformer BEFORE its state dict is handed to libgdf"""
    import diffusion_feature
    monkeypatch.setenv("FAKE_DIFFUSERS_FLUX_LAYERS", "1,1")
    df = diffusion_feature.FeatureExtractor(layer={"vit-block1-out": True}, version="flux", device="cuda:0", img_size=1024, offline_lora="/data/flux_lora",
                                            offline_lora_filename="lora.safetensors")
    calls = [c[0] for c in D.CALLS]
    assert calls.index("fuse_lora") > calls.index("load_lora_weights") > calls.index("FluxImg2ImgPipeline.from_pretrained")
    assert dict(D.CALLS)["load_lora_weights"] == dict(path="/data/flux_lora", weight_name="lora.safetensors")
    torch.manual_seed(3)
    feats = df.extract("a prompt", batch_size=1, image=_images(1, 512, seed=4), t=200)
    torch.cuda.synchronize()
    want = _flux_oracle(df.pipe, df.pipe.last_transformer_kwargs, list(feats))       # the oracle on the FUSED module's weights
    errs = {k: _rel(feats[k], want[k]) for k in feats}
    assert max(errs.values()) <= 1.0e-3, errs


def _flux_oracle(pipe, kw, ids):
    from oracle import flux_ref as FR
    c = pipe.original["transformer"].config
    arch = dict(FR.ARCH_FLUX_DEV, num_layers=c.num_layers, num_single_layers=c.num_single_layers)
    st = FR.Store({k: True for k in ids})
    f = lambda v: v.float().cpu()
    with torch.no_grad():
        FR.flux_forward(_sd(pipe.original["transformer"]), arch, f(kw["hidden_states"]), f(kw["encoder_hidden_states"]), f(kw["pooled_projections"]),
                        f(kw["timestep"]), f(kw["img_ids"]), f(kw["txt_ids"]), guidance=f(kw["guidance"]), store=st)
    return st.feats


def test_flux_real_checkpoint_branch_stock_pipeline(D, monkeypatch):
    """'flux' through a STOCK FluxImg2ImgPipeline.__call__ (true widths, depth cut to 2 + 3 blocks): bf16 checkpoint -> 'auto' (fp16 operands),
    load-time weight guard + first-forward activation range check, ONE transformer forward per extract (the stock loop is stopped), a str prompt
    expanded to the image batch."""
    import diffusion_feature
    from components.native import NativeFluxTransformer
    import conftest
    monkeypatch.setenv("FAKE_DIFFUSERS_FLUX_LAYERS", "2,3")
    layer = {"vit-block0-q": True, "vit-block1-ffn-inner": True, "vit-block1-out": True, "vit-block3-attn-out": True, "vit-block4-out": True}
    df = diffusion_feature.FeatureExtractor(layer=layer, version="flux", device="cuda:0", img_size=1024)
    pipe = df.pipe
    tr = pipe.transformer
    fp = dict(D.CALLS)["FluxImg2ImgPipeline.from_pretrained"]
    assert fp["repo"] == "black-forest-labs/FLUX.1-dev" and fp["torch_dtype"] == torch.bfloat16      # reference models.py:150-169
    assert isinstance(tr, NativeFluxTransformer) and pipe.unet is tr and tr.cfg["num_layers"] == 2 and tr.cfg["num_single_layers"] == 3
    assert tr.cfg["compute_dtype"] == "auto" and tr.fp16_cast_error <= tr.FP16_CAST_TOL and tr._range_check_sd is not None
    feats = df.extract("a macro photo of a dragonfly", batch_size=2, image=_images(2, 640, seed=7), t=300)
    torch.cuda.synchronize()
    assert pipe.transformer_calls == 1 and tr.calls == 1                 # the stock loop would have gone on to scheduler.step
    log = tr.range_check_log
    assert log and log["saturated"] == [] and log["mode_after"] == "auto" and tr._range_check_sd is None and log["tensors_scanned"] == 30
    kw = pipe.last_transformer_kwargs
    assert kw["hidden_states"].shape == (2, 4096, 64) and kw["encoder_hidden_states"].shape == (2, 512, 4096) and kw["step"] == 0
    # strength 0.3 of 28 steps: t_start = int(28 - 8.4) = 19; the transformer saw sigmas[19] of the shifted schedule
    sig = np.linspace(1.0, 1 / 28, 28); mu = D.calculate_shift(4096, 256, 4096, 0.5, 1.15)
    assert abs(kw["sigma"] - float(np.exp(mu) / (np.exp(mu) + (1 / sig[19] - 1)))) < 1e-3      # (the timestep travels as a bf16 latents-dtype tensor)
    want = _flux_oracle(pipe, kw, list(feats))
    errs = {k: _rel(feats[k], want[k]) for k in feats}
    print(f"\n[fake-diffusers flux 1024^2 B=2, 2+3 blocks, stock pipeline] hooks: {errs}")
    assert list(feats) == list(want) and max(errs.values()) <= 1.0e-3, errs
    conftest.record_margin("fake-diffusers flux 1024^2 B=2 (stock pipeline, 'auto' + range check)", max(errs, key=errs.get), max(errs.values()), 1.0e-3)
    # a second call: one more forward, no second range check
    df.extract(["one", "two"], batch_size=2, image=_images(2, 640, seed=8), t=300)
    assert pipe.transformer_calls == 2 and tr.calls == 2 and tr.range_check_log is log


def test_flux_activation_range_check_falls_back_to_bf16_pairs(D, monkeypatch):
    """a checkpoint whose ACTIVATIONS leave the fp16 range (weights all representable: the load-time guard passes) is re-loaded in
    'bfloat16x2' on its first forward, with one warning; the result equals a model that ran in 'bfloat16x2' from the start, bit for bit."""
    import diffusion_feature
    monkeypatch.setenv("FAKE_DIFFUSERS_FLUX_LAYERS", "1,1")
    layer = {"vit-block0-q": True, "vit-block1-out": True}
    real_build = D.FluxImg2ImgPipeline._build

    def build(self, repo, dt, seed, kw):
        real_build(self, repo, dt, seed, kw)
        with torch.no_grad():                                            # v = x W_v^T grows 30000-fold: sigma(v) ~ 3e4, |v| > 65504 on many elements; max |w| ~ 3e3 stays representable
            getattr(self.transformer.transformer_blocks, "0").attn.to_v.weight.mul_(30000.0)
    monkeypatch.setattr(D.FluxImg2ImgPipeline, "_build", build)
    imgs = _images(1, 512, seed=9)
    df = diffusion_feature.FeatureExtractor(layer=layer, version="flux", device="cuda:0", img_size=1024)
    tr = df.pipe.transformer
    assert tr.cfg["compute_dtype"] == "auto" and tr.fp16_cast_error <= tr.FP16_CAST_TOL
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        torch.manual_seed(5)                                             # (the stock pipeline draws its noise from the global generator)
        feats = {k: v.clone() for k, v in df.extract("p", batch_size=1, image=imgs, t=500).items()}
    msgs = [str(x.message) for x in w if "fp16 range" in str(x.message)]
    assert len(msgs) == 1 and "bfloat16x2" in msgs[0], [str(x.message) for x in w]
    assert tr.cfg["compute_dtype"] == "bfloat16x2" and tr.range_check_log["saturated"] and tr.range_check_log["mode_before"] == "auto"
    assert any(k.endswith("-v") or k.endswith("attn-out") for k, _ in tr.range_check_log["saturated"])
    monkeypatch.setenv("GDF_FLUX_DTYPE", "bfloat16x2")
    D.reset()
    df2 = diffusion_feature.FeatureExtractor(layer=layer, version="flux", device="cuda:0", img_size=1024)
    assert df2.pipe.transformer.cfg["compute_dtype"] == "bfloat16x2" and df2.pipe.transformer._range_check_sd is None
    torch.manual_seed(5)
    f2 = df2.extract("p", batch_size=1, image=imgs, t=500)
    torch.cuda.synchronize()
    for k in layer:
        assert torch.equal(feats[k], f2[k]), k


def _tree(d):
    out = {}
    for r, _, fs in os.walk(d):
        for f in fs:
            if f.endswith(".npy"):
                out[os.path.relpath(os.path.join(r, f), d)] = np.load(os.path.join(r, f))
    return out


def test_two_rank_cli_real_checkpoint_branch_broadcast(tmp_path):
    """`extract_feature.py --gpus 2` on the real-checkpoint branch: rank 0 loads the UNet from the pipeline, rank 1 is handed `unet=None`
    (components/models.py: only rank 0 reads the 5 GB checkpoint), receives the architecture descriptor and the weight arena, and its files are
    the single-process files bit for bit."""
    from PIL import Image
    (tmp_path / "imgs").mkdir()
    for i, im in enumerate(_images(4, 128, seed=11)):
        im.save(tmp_path / "imgs" / f"{i}.png")
    (tmp_path / "prompt.txt").write_text("a photo of a dog")
    (tmp_path / "layers.json").write_text(json.dumps({"up-level1-repeat2-res-out": True, "up-level2-repeat1-vit-block0-cross-q": True, "vae-out": True}))
    base = [os.path.join(ROOT, "extract_feature.py"), "--layer", str(tmp_path / "layers.json"), "--version", "1-5", "--img_size", "128",
            "--t", "100", "-b", "2", "--input_dir", str(tmp_path / "imgs" / "*.png"), "--prompt_file", str(tmp_path / "prompt.txt"),
            # a real pipeline draws the VAE sample / the noise from the global generator (reference diffusion_feature.py:371-380 passes no generator):
            # --seed makes the draws a function of (seed, batch start index), hence independent of how the images are dealt to ranks
            "--seed", "7"]
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "GDF_SYNTHETIC_WEIGHTS")}
    env.update(PYTHONPATH=FAKE + os.pathsep + env.get("PYTHONPATH", ""), HSA_ENABLE_IPC_MODE_LEGACY="0", FAKE_DIFFUSERS_LOG=str(tmp_path / "calls.jsonl"))
    r = subprocess.run([sys.executable] + base + ["--output_dir", str(tmp_path / "one")], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    one_calls = [json.loads(l) for l in open(tmp_path / "calls.jsonl")]
    assert [c["unet_given"] for c in one_calls if c["what"].endswith("from_pretrained")] == [False]
    os.remove(tmp_path / "calls.jsonl")
    r = subprocess.run([sys.executable] + base + ["--gpus", "2", "--output_dir", str(tmp_path / "two")], env=dict(env, GDF_SHARE_GPU="1"),
                       capture_output=True, text=True, timeout=1200)
    assert r.returncode == 0, r.stderr[-3000:]
    calls = {c["rank"]: c for c in (json.loads(l) for l in open(tmp_path / "calls.jsonl")) if c["what"].endswith("from_pretrained")}
    assert calls["0"]["unet_given"] is False and calls["1"]["unet_given"] is True and calls["1"]["unet_is_none"] is True
    one, two = _tree(tmp_path / "one"), _tree(tmp_path / "two")
    assert sorted(one) == sorted(two) and len(one) == 3 * 4
    for k in one:
        assert np.array_equal(one[k].view(np.uint16), two[k].view(np.uint16)), k
