"""GPU parity of the VAE-encoder + sampling + noise-add stage (include/gdf_vae.h) against the CPU
oracle (oracle/vae_ref.py, pinned block-wise to the reference's resnet / downsample / attention modules).
Stated tolerance: relative L2 error <= 3e-3 on the latents (fp16 MFMA operands, fp32 accumulate)."""
import pytest
import torch

from helpers import rel_l2
from oracle import vae_ref as VR

pytestmark = pytest.mark.gpu


def _ops():
    from ops_binding import P, lib, ok, stream
    return lib(), P, ok, stream


def test_softmax_rows():
    L, P, ok, stream = _ops()
    import ctypes
    L.gdf_op_softmax_rows.restype = ctypes.c_int
    for n in (64, 1024, 16384):
        x = (torch.randn(37, n, device="cuda") * 20).half()
        y = x.clone()
        ok(L.gdf_op_softmax_rows(P(y), n, 37, n, ctypes.c_float(0.0442), stream()), L)
        ref = torch.softmax(x.float() * 0.0442, -1)
        assert rel_l2(y, ref) < 2e-3 and torch.allclose(y.float().sum(-1), torch.ones(37, device="cuda"), atol=5e-3)


@pytest.mark.parametrize("channels,img,batch", [((64, 128, 128), 64, 2), ((64, 128, 256, 256), 128, 3)])
def test_vae_encode_matches_oracle(channels, img, batch):
    from components.native import NativeVAEEncoder
    arch = VR.tiny_arch(channels)
    P = VR.synth_params(arch, seed=0)
    g = torch.Generator().manual_seed(1)
    image = (torch.rand(batch, 3, img, img, generator=g) * 2 - 1).half().float()
    lat = img >> (len(channels) - 1)
    eps = torch.randn(batch, 4, lat, lat, generator=g).half().float()
    noise = torch.randn(batch, 4, lat, lat, generator=g).half().float()
    cfg = dict(in_channels=3, latent_channels=4, block_out_channels=channels, layers_per_block=2, use_quant_conv=1)
    enc = NativeVAEEncoder(cfg, device="cuda:0")
    enc.load_vae_state_dict({k: v.half() for k, v in P.items()})
    assert enc.ready()
    for kw in (dict(eps=None, noise=None, scaling_factor=1.0, noise_a=1.0, noise_b=0.0, input_scale=1.0),          # posterior mode
               dict(eps=eps, noise=noise, scaling_factor=0.13025, noise_a=1.0, noise_b=0.7, input_scale=0.82),      # Euler (SDXL)
               dict(eps=eps, noise=noise, scaling_factor=0.18215, noise_a=0.95, noise_b=0.31, input_scale=1.0)):   # DDPM (SD1.5)
        ref = VR.prepare_latents(P, arch, image, kw["eps"], kw["noise"], kw["scaling_factor"], kw["noise_a"], kw["noise_b"],
                                 kw["input_scale"])
        got = enc.encode(image, **kw)
        torch.cuda.synchronize()
        assert got.shape == ref.shape and got.dtype == torch.float16
        assert rel_l2(got, ref) < 3e-3, (kw, rel_l2(got, ref))


def test_vae_encode_range_beyond_fp16():
    """Range safety: the reference upcasts the SDXL VAE to fp32 because its activations exceed the fp16 range.  Here the
    residual stream is pushed to ~1e5 (> 65504) by the first resnet (conv2 weights x 1e5: fp16-representable weights, K = 576
    products per output) and stays there through every later resnet, the downsamplers and the mid attention: the fp32 masters + 2^-6-scaled fp16 images (csrc/vae.cpp) must reproduce the fp32
    oracle at the normal tolerance — no inf / NaN, no saturation."""
    from components.native import NativeVAEEncoder
    channels = (64, 128, 128)
    arch = VR.tiny_arch(channels)
    P = VR.synth_params(arch, seed=0)
    k = "encoder.down_blocks.0.resnets.0.conv2."
    P[k + "weight"] = (P[k + "weight"] * 1.0e5).half().float()
    assert torch.isfinite(P[k + "weight"]).all()
    P[k + "bias"] = P[k + "bias"] * 1.0e5
    g = torch.Generator().manual_seed(1)
    image = (torch.rand(2, 3, 64, 64, generator=g) * 2 - 1).half().float()
    cfg = dict(in_channels=3, latent_channels=4, block_out_channels=channels, layers_per_block=2, use_quant_conv=1)
    enc = NativeVAEEncoder(cfg, device="cuda:0")
    enc.load_vae_state_dict({k: v.float() for k, v in P.items()})           # fp32 source: 2e5 x weights do not fit fp16 themselves
    import torch.nn.functional as F
    x0 = F.conv2d(image, P["encoder.conv_in.weight"], P["encoder.conv_in.bias"], padding=1)
    r = "encoder.down_blocks.0.resnets.0."
    h = F.conv2d(F.silu(F.group_norm(x0, 32, P[r + "norm1.weight"], P[r + "norm1.bias"], 1e-6)), P[r + "conv1.weight"], P[r + "conv1.bias"], padding=1)
    h = F.conv2d(F.silu(F.group_norm(h, 32, P[r + "norm2.weight"], P[r + "norm2.bias"], 1e-6)), P[r + "conv2.weight"], P[r + "conv2.bias"], padding=1)
    assert float((x0 + h).abs().max()) > 65504 * 2                           # the stream really leaves the fp16 range
    ref = VR.prepare_latents(P, arch, image, None, None, 1.0, 1.0, 0.0, 1.0)
    got = enc.encode(image, eps=None, noise=None, scaling_factor=1.0)
    torch.cuda.synchronize()
    assert torch.isfinite(got.float()).all()
    assert rel_l2(got, ref) < 3e-3, rel_l2(got, ref)


# ---- `vae-out`: scheduler step + AutoencoderKL decoder (include/gdf_vae.h, reference diffusion_feature.py:60, :477-485) ----
@pytest.mark.parametrize("channels,lat,batch", [((64, 128, 128), 16, 2), ((64, 128, 256, 256), 16, 3)])
def test_vae_decode_matches_oracle(channels, lat, batch):
    """decode((c_sample * latents + c_eps * noise_pred) / scaling_factor) against oracle/vae_ref.py: plain decode, an Euler step and
    a PNDM-style step.  Stated tolerance: relative L2 <= 3e-3 on the image (shrunken widths; 1e-3 at full size, test_gpu_fullsize)."""
    from components.native import NativeVAEDecoder
    arch = VR.tiny_arch(channels)
    P = VR.synth_dec_params(arch, seed=0)
    g = torch.Generator().manual_seed(1)
    z = torch.randn(batch, 4, lat, lat, generator=g).half().float()
    eps = torch.randn(batch, 4, lat, lat, generator=g).half().float()
    cfg = dict(in_channels=3, latent_channels=4, block_out_channels=channels, layers_per_block=2, use_quant_conv=1)
    dec = NativeVAEDecoder(cfg, device="cuda:0")
    assert dec.param_shapes() == {k: tuple(v) for k, v in VR.dec_param_shapes(arch).items()}
    dec.load_vae_state_dict({k: v.half() for k, v in P.items()})
    assert dec.ready()
    f = 1 << (len(channels) - 1)
    for (a, b, sf, with_eps) in ((1.0, 0.0, 1.0, False), (1.0, -0.35, 0.13025, True), (1.02, -0.21, 0.18215, True)):
        ref = VR.vae_out(P, arch, z, eps if with_eps else torch.zeros_like(z), a, b if with_eps else 0.0, sf)
        got = dec.decode(z, eps if with_eps else None, c_sample=a, c_eps=b, scaling_factor=sf)
        torch.cuda.synchronize()
        assert tuple(got.shape) == tuple(ref.shape) == (batch, 3, lat * f, lat * f) and got.dtype == torch.float16
        e = rel_l2(got, ref)
        assert e < 3e-3, ((a, b, sf), e)


def test_vae_decode_range_beyond_fp16():
    """The decoder stream pushed past the fp16 range (conv2 of the first mid resnet x 1e5) must still match the fp32 oracle: fp32
    masters + 2^-6-scaled fp16 images as in the encoder."""
    from components.native import NativeVAEDecoder
    channels = (64, 128, 128)
    arch = VR.tiny_arch(channels)
    P = VR.synth_dec_params(arch, seed=0)
    k = "decoder.mid_block.resnets.0.conv2."
    P[k + "weight"] = (P[k + "weight"] * 1.0e5).half().float()
    P[k + "bias"] = P[k + "bias"] * 1.0e5
    z = torch.randn(2, 4, 16, 16, generator=torch.Generator().manual_seed(1)).half().float()
    cfg = dict(in_channels=3, latent_channels=4, block_out_channels=channels, layers_per_block=2, use_quant_conv=1)
    dec = NativeVAEDecoder(cfg, device="cuda:0")
    dec.load_vae_state_dict({k: v.float() for k, v in P.items()})
    ref = VR.decode(P, arch, z)
    got = dec.decode(z, None, scaling_factor=1.0)
    torch.cuda.synchronize()
    assert torch.isfinite(got.float()).all()
    assert rel_l2(got, ref) < 3e-3, rel_l2(got, ref)


@pytest.mark.parametrize("version,img", [("1-5", 256), ("xl", 256)])
def test_feature_extractor_vae_out(version, img, monkeypatch):
    """FeatureExtractor(layer={'vae-out': True, ...}) returns the decoded image next to the hooks (reference :60, :477-485): it equals
    the native decoder applied to (latents, noise_pred) with the scheduler's first-step coefficients, which in turn are the oracle's
    PNDM / Euler formulas."""
    monkeypatch.setenv("GDF_SYNTHETIC_WEIGHTS", "1")
    import diffusion_feature
    from components.models import native_vae_decoder, scheduler_step_scalars
    hook = "up-level1-repeat1-vit-block0-cross-q" if version == "1-5" else "up-level1-repeat0-vit-block0-out"
    df = diffusion_feature.FeatureExtractor(layer={hook: True, "vae-out": True}, version=version, img_size=img, device='cuda')
    assert df.store_vae_output
    prompt = df.encode_prompt('a photo of a cat')
    lat = torch.randn(2, 4, img // 8, img // 8, generator=torch.Generator().manual_seed(0)).half()
    feats = df.extract(prompt, batch_size=2, image=lat, image_type='latents', t=100)
    assert list(feats.keys()) == [hook, "vae-out"]
    out = feats["vae-out"]
    assert tuple(out.shape) == (2, 3, img, img) and out.dtype == torch.float16 and torch.isfinite(out.float()).all()
    # the same numbers from the pieces
    import copy
    s2 = copy.deepcopy(df.scheduler_backup); s2.set_timesteps(1000, device='cuda')
    ts, _ = df.pipe.get_timesteps(1000, 100 / 1000, 'cuda')
    a, b = scheduler_step_scalars(s2, ts[:1])
    ti = int(ts[0])
    if version == "1-5":
        ra, rb = VR.pndm_first_step_scalars(s2.alphas_cumprod, ti, ti - 1)
    else:
        ac = s2.alphas_cumprod
        sig = lambda i: float(((1 - ac[i]) / ac[i]) ** 0.5)
        ra, rb = VR.euler_step_scalars(sig(ti), sig(ti - 1))
    assert abs(a - ra) < 1e-6 and abs(b - rb) < 1e-6
    unet = df.pipe.unet
    lin = df.pipe.scheduler.scale_model_input(lat.cuda(), ts[:1])
    emb, _, pooled, _ = prompt
    kw = {}
    if version == "xl":
        kw = dict(text_embeds=pooled.repeat(2, 1, 1).squeeze(1).cuda(),
                  time_ids=torch.tensor([[img, img, 0, 0, img, img]], dtype=torch.float32).repeat(2, 1).cuda())
    # (the same id list as the extractor's: 'vae-out' is no UNet hook, but the automatic operand-plan selection counts it like `unet-out`)
    noise, _ = unet.forward_raw(lin, ts[:1], emb.repeat(2, 1, 1).cuda(), kw.get("text_embeds"), kw.get("time_ids"), hook_ids=[hook, "vae-out"],
                                shared_ctx=True)
    img2 = native_vae_decoder(df.pipe, 'cuda').decode(lat.cuda(), noise, c_sample=a, c_eps=b,
                                                       scaling_factor=float(df.pipe.vae.config.scaling_factor))
    torch.cuda.synchronize()
    assert torch.equal(img2, out)


def test_vae_groupnorm_statistics_from_conv_epilogue(monkeypatch):
    """Round 4: the 3x3 convs of the VAE op programs emit the per-channel GroupNorm partial sums of the image they store
    (GemmParams::gn_partial), so the consuming GroupNorm runs `gn_finalize` instead of a statistics pass over the tensor.
    Same results as the separate pass (GDF_VAE_GN_EPI=0) and as the fp32 oracle; the op program must actually use it."""
    from components.native import NativeVAEEncoder
    channels, img, batch = (64, 128, 256, 256), 256, 2          # level 0: 256 x 256 pixels = 1024 slabs of 64 rows per sample (folded to 128)
    arch = VR.tiny_arch(channels)
    P = VR.synth_params(arch, seed=0)
    g = torch.Generator().manual_seed(3)
    image = (torch.rand(batch, 3, img, img, generator=g) * 2 - 1).half().float()
    cfg = dict(in_channels=3, latent_channels=4, block_out_channels=channels, layers_per_block=2, use_quant_conv=1)
    kw = dict(eps=None, noise=None, scaling_factor=1.0, noise_a=1.0, noise_b=0.0, input_scale=1.0)
    outs, names = {}, {}
    for flag in ("0", "1"):
        monkeypatch.setenv("GDF_VAE_GN_EPI", flag)
        enc = NativeVAEEncoder(cfg, device="cuda:0")
        enc.load_vae_state_dict({k: v.half() for k, v in P.items()})
        out, prof = enc.encode(image, profile=True, **kw)
        torch.cuda.synchronize()
        outs[flag] = out.float().cpu()
        names[flag] = [n for n, *_ in prof]
    assert "gn_finalize" not in names["0"] and names["0"].count("gn_stats") > 4
    assert names["1"].count("gn_finalize") >= 6, names["1"]                 # conv_in, both convs of the level-0 / level-1 resnets, the downsamplers
    ref = VR.prepare_latents(P, arch, image, None, None, 1.0, 1.0, 0.0, 1.0)
    e1, e0, d = rel_l2(outs["1"], ref), rel_l2(outs["0"], ref), rel_l2(outs["1"], outs["0"])
    print(f"[vae gn-epilogue] vs oracle: epilogue statistics {e1:.2e}, separate pass {e0:.2e}; between the two {d:.2e}")
    assert e1 < 3e-3 and e0 < 3e-3
    assert d < 2e-3            # the epilogue sums the fp32 values before the fp16 rounding of the stored image, the separate pass the rounded image


@pytest.mark.parametrize("img,batch", [(320, 1), (320, 3), (192, 2)])
def test_vae_gn_epilogue_ragged_last_tile(img, batch, monkeypatch):
    """image sizes that are odd multiples of 64 give conv levels whose row count M is a multiple of 64 but NOT of the
    128- / 256-row workgroup tile (320^2, batch 1: the 40 x 40 level has M = 1600 = 12.5 tiles of 128).  The waves of the last tile whose
    64 rows start beyond M own no statistics slab and must not store one (they used to write N * 8 bytes past the M / 64 slabs).
    Encode AND decode with the statistics epilogue must equal the separate-pass build and the fp32 oracle."""
    from components.native import NativeVAEEncoder, NativeVAEDecoder
    channels = (64, 128, 256, 256)
    arch = VR.tiny_arch(channels)
    P, PD = VR.synth_params(arch, seed=0), VR.synth_dec_params(arch, seed=0)
    g = torch.Generator().manual_seed(5)
    image = (torch.rand(batch, 3, img, img, generator=g) * 2 - 1).half().float()
    lat = img // 8
    z = torch.randn(batch, 4, lat, lat, generator=g).half().float()
    cfg = dict(in_channels=3, latent_channels=4, block_out_channels=channels, layers_per_block=2, use_quant_conv=1)
    kw = dict(eps=None, noise=None, scaling_factor=1.0, noise_a=1.0, noise_b=0.0, input_scale=1.0)
    enc_out, dec_out = {}, {}
    for flag in ("0", "1"):
        monkeypatch.setenv("GDF_VAE_GN_EPI", flag)
        enc = NativeVAEEncoder(cfg, device="cuda:0")
        enc.load_vae_state_dict({k: v.half() for k, v in P.items()})
        dec = NativeVAEDecoder(cfg, device="cuda:0")
        dec.load_vae_state_dict({k: v.half() for k, v in PD.items()})
        for _ in range(2):                                   # twice: a stray store of the first run would corrupt the second one's inputs
            e = enc.encode(image, **kw)
            d = dec.decode(z, None, scaling_factor=1.0)
        torch.cuda.synchronize()
        enc_out[flag], dec_out[flag] = e.float().cpu(), d.float().cpu()
    ref_e = VR.prepare_latents(P, arch, image, None, None, 1.0, 1.0, 0.0, 1.0)
    ref_d = VR.decode(PD, arch, z)
    for flag in ("0", "1"):
        assert rel_l2(enc_out[flag], ref_e) < 3e-3 and rel_l2(dec_out[flag], ref_d) < 3e-3, (flag, rel_l2(enc_out[flag], ref_e), rel_l2(dec_out[flag], ref_d))
    assert rel_l2(enc_out["1"], enc_out["0"]) < 2e-3 and rel_l2(dec_out["1"], dec_out["0"]) < 2e-3
