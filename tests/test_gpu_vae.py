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
