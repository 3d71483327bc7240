"""CPU ORACLE (test infrastructure only) — fp32 restatement of the step immediately BEFORE the hot path
(SURVEY.md §8f rank 1): VAE encode + latent sampling + scheduler noise-add + scale_model_input, i.e. what
`self.pipe.prepare_latents(image, latent_timestep, ...)` and `scheduler.scale_model_input` do at
/root/reference/feature/diffusion_feature.py:371-380, :405-406.

This file is a CHECKER (tests/, smoke(), bench cpu_baseline only); the product path never imports it.

AutoencoderKL / Encoder / DownEncoderBlock2D / UNetMidBlock2D / DiagonalGaussianDistribution and the schedulers are
un-vendored diffusers==0.32.2 (restated from the published algorithm => "parity unpinned" for the wiring).  The blocks
they are made of ARE in the reference tree and pin this file through tests/golden/vae_*.npz
(tests/golden/gen_golden_vae.py runs the reference's own modules):
  ResnetBlock2D(temb_channels=None, eps=1e-6)      feature/diffusers/models/resnet.py:189-379
  Downsample2D(padding=0): F.pad (0,1,0,1) + conv stride 2   feature/diffusers/models/downsampling.py:132-152
  Attention(heads=1, dim_head=C, norm_num_groups=32, residual_connection=True, bias=True) + AttnProcessor2_0 on a
  4-D input                                         feature/diffusers/models/attention_processor.py:50-297, 3244-3331

`vae-out` (the optional last id of the reference's layer grammar, diffusion_feature.py:60, :477-485): one scheduler step on the
un-scaled latents followed by `vae.decode(latents / scaling_factor)`.  The decoder half (Decoder / UNetMidBlock2D /
UpDecoderBlock2D, un-vendored like the encoder) is built from the same in-tree blocks plus
  Upsample2D(use_conv=True): nearest x2 + conv3x3      feature/diffusers/models/upsampling.py:142-195 (golden: upsample.npz)
and the first-call arithmetic of PNDMScheduler.step / EulerDiscreteScheduler.step is restated from the published algorithm
(un-vendored => "parity unpinned" for the wiring and the scheduler formulas).
"""
import math
from collections import OrderedDict

import torch
import torch.nn.functional as F

# AutoencoderKL config of the SD1.5 / SDXL VAEs (vae/config.json of the checkpoints the reference downloads)
ARCH_SD_VAE = dict(in_channels=3, latent_channels=4, block_out_channels=(128, 256, 512, 512), layers_per_block=2,
                   norm_num_groups=32, use_quant_conv=True)


def tiny_arch(channels=(64, 128, 128)):
    a = dict(ARCH_SD_VAE)
    a["block_out_channels"] = tuple(channels)
    return a


def param_shapes(arch):
    """`vae.state_dict()` names of the encoder half (+ quant_conv)."""
    boc = arch["block_out_channels"]; nl = arch["layers_per_block"]; L = len(boc)
    S = OrderedDict()

    def conv(n, co, ci, k):
        S[n + ".weight"] = (co, ci, k, k); S[n + ".bias"] = (co,)

    def norm(n, c):
        S[n + ".weight"] = (c,); S[n + ".bias"] = (c,)

    def res(p, ci, co):
        norm(p + ".norm1", ci); conv(p + ".conv1", co, ci, 3); norm(p + ".norm2", co); conv(p + ".conv2", co, co, 3)
        if ci != co:
            conv(p + ".conv_shortcut", co, ci, 1)

    conv("encoder.conv_in", boc[0], arch["in_channels"], 3)
    ci = boc[0]
    for lv in range(L):
        for r in range(nl):
            res(f"encoder.down_blocks.{lv}.resnets.{r}", ci, boc[lv]); ci = boc[lv]
        if lv != L - 1:
            conv(f"encoder.down_blocks.{lv}.downsamplers.0.conv", boc[lv], boc[lv], 3)
    c = boc[-1]
    res("encoder.mid_block.resnets.0", c, c)
    a = "encoder.mid_block.attentions.0"
    norm(a + ".group_norm", c)
    for n in ("to_q", "to_k", "to_v", "to_out.0"):
        S[f"{a}.{n}.weight"] = (c, c); S[f"{a}.{n}.bias"] = (c,)
    res("encoder.mid_block.resnets.1", c, c)
    norm("encoder.conv_norm_out", c)
    conv("encoder.conv_out", 2 * arch["latent_channels"], c, 3)
    if arch["use_quant_conv"]:
        conv("quant_conv", 2 * arch["latent_channels"], 2 * arch["latent_channels"], 1)
    return S


def synth_params(arch, seed=0, dtype=torch.float32):
    g = torch.Generator().manual_seed(seed)
    P = OrderedDict()
    for name, shape in param_shapes(arch).items():
        is_norm = "norm" in name
        if name.endswith(".weight") and not is_norm:
            fan_in = 1
            for s in shape[1:]:
                fan_in *= s
            w = torch.randn(shape, generator=g) / math.sqrt(fan_in)
        elif name.endswith(".weight"):
            w = 1.0 + 0.1 * torch.randn(shape, generator=g)
        elif is_norm:
            w = 0.1 * torch.randn(shape, generator=g)
        else:
            w = 0.05 * torch.randn(shape, generator=g)
        P[name] = w.half().to(dtype)
    return P


def resnet_block(P, pfx, x, eps=1e-6, groups=32):
    """ResnetBlock2D.forward with temb=None (resnet.py:320-379; the time-embedding branch :343-350 is skipped)."""
    h = F.silu(F.group_norm(x, groups, P[pfx + ".norm1.weight"], P[pfx + ".norm1.bias"], eps))
    h = F.conv2d(h, P[pfx + ".conv1.weight"], P[pfx + ".conv1.bias"], padding=1)
    h = F.silu(F.group_norm(h, groups, P[pfx + ".norm2.weight"], P[pfx + ".norm2.bias"], eps))
    h = F.conv2d(h, P[pfx + ".conv2.weight"], P[pfx + ".conv2.bias"], padding=1)
    if (pfx + ".conv_shortcut.weight") in P:
        x = F.conv2d(x, P[pfx + ".conv_shortcut.weight"], P[pfx + ".conv_shortcut.bias"])
    return x + h


def downsample_pad0(P, pfx, x):
    """Downsample2D.forward, padding == 0 (downsampling.py:141-143): pad right/bottom by one, conv stride 2."""
    x = F.pad(x, (0, 1, 0, 1), mode="constant", value=0)
    return F.conv2d(x, P[pfx + ".conv.weight"], P[pfx + ".conv.bias"], stride=2)


def mid_attention(P, pfx, x, groups=32, eps=1e-6):
    """Attention (single head, dim_head = C) + AttnProcessor2_0 on (B,C,H,W) (attention_processor.py:3244-3331:
    view -> group_norm -> q,k,v (bias) -> SDPA -> to_out[0] -> reshape -> + residual)."""
    b, c, hh, ww = x.shape
    t = F.group_norm(x, groups, P[pfx + ".group_norm.weight"], P[pfx + ".group_norm.bias"], eps)
    t = t.view(b, c, hh * ww).transpose(1, 2)
    q = F.linear(t, P[pfx + ".to_q.weight"], P[pfx + ".to_q.bias"])
    k = F.linear(t, P[pfx + ".to_k.weight"], P[pfx + ".to_k.bias"])
    v = F.linear(t, P[pfx + ".to_v.weight"], P[pfx + ".to_v.bias"])
    o = F.scaled_dot_product_attention(q[:, None], k[:, None], v[:, None])[:, 0]
    o = F.linear(o, P[pfx + ".to_out.0.weight"], P[pfx + ".to_out.0.bias"])
    return o.transpose(1, 2).reshape(b, c, hh, ww) + x


def encoder_moments(P, arch, image):
    """AutoencoderKL.encode up to the posterior parameters: Encoder.forward + quant_conv -> (mean, logvar)."""
    boc = arch["block_out_channels"]; nl = arch["layers_per_block"]; L = len(boc)
    h = F.conv2d(image.float(), P["encoder.conv_in.weight"], P["encoder.conv_in.bias"], padding=1)
    for lv in range(L):
        for r in range(nl):
            h = resnet_block(P, f"encoder.down_blocks.{lv}.resnets.{r}", h)
        if lv != L - 1:
            h = downsample_pad0(P, f"encoder.down_blocks.{lv}.downsamplers.0", h)
    h = resnet_block(P, "encoder.mid_block.resnets.0", h)
    h = mid_attention(P, "encoder.mid_block.attentions.0", h)
    h = resnet_block(P, "encoder.mid_block.resnets.1", h)
    h = F.silu(F.group_norm(h, 32, P["encoder.conv_norm_out.weight"], P["encoder.conv_norm_out.bias"], 1e-6))
    h = F.conv2d(h, P["encoder.conv_out.weight"], P["encoder.conv_out.bias"], padding=1)
    if arch["use_quant_conv"]:
        h = F.conv2d(h, P["quant_conv.weight"], P["quant_conv.bias"])
    mean, logvar = h.chunk(2, dim=1)
    return mean, logvar.clamp(-30.0, 20.0)                        # DiagonalGaussianDistribution.__init__


def prepare_latents(P, arch, image, eps, noise, scaling_factor, noise_a, noise_b, input_scale=1.0):
    """StableDiffusion(XL)Img2ImgPipeline.prepare_latents + scheduler.scale_model_input
    (call sites diffusion_feature.py:371-380, :405-406):
        z        = mean + exp(0.5 logvar) * eps                (latent_dist.sample; eps=None -> mode)
        latents  = scaling_factor * z
        noisy    = noise_a * latents + noise_b * noise          (add_noise: DDPM sqrt(ac), sqrt(1-ac); Euler 1, sigma)
        return input_scale * noisy                              (scale_model_input: 1 or 1/sqrt(sigma^2+1))"""
    mean, logvar = encoder_moments(P, arch, image)
    z = mean if eps is None else mean + torch.exp(0.5 * logvar) * eps.float()
    lat = scaling_factor * z
    if noise is not None:
        lat = noise_a * lat + noise_b * noise.float()
    return input_scale * lat


def flops_per_image(arch, img):
    """2*MACs of the encoder convs / linears / attention at img x img input."""
    boc = arch["block_out_channels"]; nl = arch["layers_per_block"]; L = len(boc)
    fl = 0.0
    hw = img * img
    fl += 2.0 * hw * arch["in_channels"] * boc[0] * 9
    ci = boc[0]
    for lv in range(L):
        for _ in range(nl):
            co = boc[lv]
            fl += 2.0 * hw * 9 * (ci * co + co * co) + (2.0 * hw * ci * co if ci != co else 0.0)
            ci = co
        if lv != L - 1:
            hw //= 4
            fl += 2.0 * hw * 9 * boc[lv] * boc[lv]
    c = boc[-1]
    fl += 2 * 2.0 * hw * 9 * 2 * c * c                  # two mid resnets
    fl += 4 * 2.0 * hw * c * c + 4.0 * hw * hw * c      # q,k,v,out + QK^T + PV
    fl += 2.0 * hw * 9 * c * 2 * arch["latent_channels"]
    return fl


# ----------------------------------------------------------------------------------------------------------------------
# `vae-out`: scheduler.step + AutoencoderKL.decode  (call site /root/reference/feature/diffusion_feature.py:477-485)
# ----------------------------------------------------------------------------------------------------------------------
def dec_param_shapes(arch):
    """`vae.state_dict()` names of the decoder half (+ post_quant_conv)."""
    boc = arch["block_out_channels"]; nl = arch["layers_per_block"]; L = len(boc)
    rev = tuple(reversed(boc))
    S = OrderedDict()

    def conv(n, co, ci, k):
        S[n + ".weight"] = (co, ci, k, k); S[n + ".bias"] = (co,)

    def norm(n, c):
        S[n + ".weight"] = (c,); S[n + ".bias"] = (c,)

    def res(p, ci, co):
        norm(p + ".norm1", ci); conv(p + ".conv1", co, ci, 3); norm(p + ".norm2", co); conv(p + ".conv2", co, co, 3)
        if ci != co:
            conv(p + ".conv_shortcut", co, ci, 1)

    if arch.get("use_post_quant_conv", arch["use_quant_conv"]):
        conv("post_quant_conv", arch["latent_channels"], arch["latent_channels"], 1)
    c = rev[0]
    conv("decoder.conv_in", c, arch["latent_channels"], 3)
    res("decoder.mid_block.resnets.0", c, c)
    a = "decoder.mid_block.attentions.0"
    norm(a + ".group_norm", c)
    for n in ("to_q", "to_k", "to_v", "to_out.0"):
        S[f"{a}.{n}.weight"] = (c, c); S[f"{a}.{n}.bias"] = (c,)
    res("decoder.mid_block.resnets.1", c, c)
    ci = c
    for i in range(L):
        for r in range(nl + 1):
            res(f"decoder.up_blocks.{i}.resnets.{r}", ci, rev[i]); ci = rev[i]
        if i != L - 1:
            conv(f"decoder.up_blocks.{i}.upsamplers.0.conv", rev[i], rev[i], 3)
    norm("decoder.conv_norm_out", boc[0])
    conv("decoder.conv_out", arch["in_channels"], boc[0], 3)
    return S


def synth_dec_params(arch, seed=0, dtype=torch.float32):
    g = torch.Generator().manual_seed(seed)
    P = OrderedDict()
    for name, shape in dec_param_shapes(arch).items():
        is_norm = "norm" in name
        if name.endswith(".weight") and not is_norm:
            fan_in = 1
            for s_ in shape[1:]:
                fan_in *= s_
            w = torch.randn(shape, generator=g) / math.sqrt(fan_in)
        elif name.endswith(".weight"):
            w = 1.0 + 0.1 * torch.randn(shape, generator=g)
        elif is_norm:
            w = 0.1 * torch.randn(shape, generator=g)
        else:
            w = 0.05 * torch.randn(shape, generator=g)
        P[name] = w.half().to(dtype)
    return P


def upsample_conv(P, pfx, x):
    """Upsample2D.forward (upsampling.py:142-195): F.interpolate(scale 2, nearest) (:176-177) then conv3x3 (:190)."""
    y = F.interpolate(x, scale_factor=2.0, mode="nearest")
    return F.conv2d(y, P[pfx + ".conv.weight"], P[pfx + ".conv.bias"], padding=1)


def decode(P, arch, z):
    """AutoencoderKL.decode: post_quant_conv -> Decoder.forward (conv_in, UNetMidBlock2D, UpDecoderBlock2D x L with
    layers_per_block + 1 resnets each and an Upsample2D on all but the last, conv_norm_out + SiLU, conv_out)."""
    boc = arch["block_out_channels"]; nl = arch["layers_per_block"]; L = len(boc)
    h = z.float()
    if "post_quant_conv.weight" in P:
        h = F.conv2d(h, P["post_quant_conv.weight"], P["post_quant_conv.bias"])
    h = F.conv2d(h, P["decoder.conv_in.weight"], P["decoder.conv_in.bias"], padding=1)
    h = resnet_block(P, "decoder.mid_block.resnets.0", h)
    h = mid_attention(P, "decoder.mid_block.attentions.0", h)
    h = resnet_block(P, "decoder.mid_block.resnets.1", h)
    for i in range(L):
        for r in range(nl + 1):
            h = resnet_block(P, f"decoder.up_blocks.{i}.resnets.{r}", h)
        if i != L - 1:
            h = upsample_conv(P, f"decoder.up_blocks.{i}.upsamplers.0", h)
    h = F.silu(F.group_norm(h, 32, P["decoder.conv_norm_out.weight"], P["decoder.conv_norm_out.bias"], 1e-6))
    return F.conv2d(h, P["decoder.conv_out.weight"], P["decoder.conv_out.bias"], padding=1)


def pndm_first_step_scalars(alphas_cumprod, t, t_prev):
    """PNDMScheduler.step (skip_prk_steps: step_plms), FIRST call after set_timesteps (counter 0, empty `ets`): the model
    output is used as is and prev_sample = _get_prev_sample(sample, t, t_prev, model_output) (PNDM eq. 9, epsilon prediction):
        prev = sqrt(a_prev / a_t) * sample - (a_prev - a_t) * eps / (a_t * sqrt(1 - a_prev) + sqrt(a_t * (1 - a_t) * a_prev))
    Returns (c_sample, c_eps).  a_prev = final_alpha_cumprod (alphas_cumprod[0], set_alpha_to_one=False) when t_prev < 0."""
    a_t = float(alphas_cumprod[int(t)])
    a_p = float(alphas_cumprod[int(t_prev)]) if t_prev >= 0 else float(alphas_cumprod[0])
    denom = a_t * math.sqrt(1.0 - a_p) + math.sqrt(a_t * (1.0 - a_t) * a_p)
    return math.sqrt(a_p / a_t), -(a_p - a_t) / denom


def euler_step_scalars(sigma, sigma_next):
    """EulerDiscreteScheduler.step (epsilon prediction, s_churn = 0): pred_x0 = sample - sigma * eps, derivative = eps,
    prev = sample + (sigma_next - sigma) * eps.  Returns (c_sample, c_eps)."""
    return 1.0, float(sigma_next) - float(sigma)


def vae_out(P, arch, latents, noise_pred, c_sample, c_eps, scaling_factor):
    """diffusion_feature.py:477-485: latents = scheduler.step(noise_pred, t, latents)[0]; vae.decode(latents / scaling_factor)[0]."""
    z = (c_sample * latents.float() + c_eps * noise_pred.float()) / scaling_factor
    return decode(P, arch, z)


def dec_flops_per_image(arch, img):
    """2*MACs of the decoder convs / linears / attention for an img x img output."""
    boc = arch["block_out_channels"]; nl = arch["layers_per_block"]; L = len(boc)
    rev = tuple(reversed(boc))
    hw = (img >> (L - 1)) ** 2
    c = rev[0]
    fl = 2.0 * hw * arch["latent_channels"] * c * 9
    fl += 2 * 2.0 * hw * 9 * 2 * c * c + 4 * 2.0 * hw * c * c + 4.0 * hw * hw * c
    ci = c
    for i in range(L):
        for _ in range(nl + 1):
            co = rev[i]
            fl += 2.0 * hw * 9 * (ci * co + co * co) + (2.0 * hw * ci * co if ci != co else 0.0)
            ci = co
        if i != L - 1:
            hw *= 4
            fl += 2.0 * hw * 9 * rev[i] * rev[i]
    fl += 2.0 * hw * 9 * boc[0] * arch["in_channels"]
    return fl
