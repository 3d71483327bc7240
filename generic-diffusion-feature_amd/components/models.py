"""Model registry of the native path — same entry point and version/dtype strings as the reference's
feature/components/models.py::get_diffusion_model (:10-175).

What is native here is the DENOISER (`pipe.unet` -> components.native.NativeUNet, libgdf.so).  Everything
upstream of the hot path (text encoders, VAE encoder, scheduler) is, as in the reference, whatever the
diffusers pipeline provides.  Offline (no diffusers, no checkpoints — the situation of the build and
benchmark boxes) `GDF_SYNTHETIC_WEIGHTS=1` selects a SyntheticPipe: seeded random UNet weights of the true
architecture plus deterministic stand-ins for prompt encoding / latent preparation, so the hot path can be
exercised and measured end to end.
"""
import hashlib
import math
import os
import types

import torch

from .native import ARCH_CONFIGS, NativeUNet, config_from_diffusers

# version -> (HF repo id, pipeline class name) exactly as the reference selects them (models.py:18-70)
_HF = {
    "1-5": ("runwayml/stable-diffusion-v1-5", "StableDiffusionImg2ImgPipeline"),
    "xl": ("stabilityai/stable-diffusion-xl-base-1.0", "StableDiffusionXLImg2ImgPipeline"),
    "pgv2": ("playgroundai/playground-v2-1024px-aesthetic", "StableDiffusionXLImg2ImgPipeline"),
}
_LATER = ("2-1", "pixart-sigma", "pixart-sigma-512", "pixart-alpha", "flux", "if", "hunyuan")


def _parse_dtype(dtype):
    if dtype == 'float32':
        return torch.float32
    if dtype == 'float16':
        return torch.float16
    raise NotImplementedError                                    # reference models.py:11-16


class _Scheduler:
    """Minimal noise schedule (scaled-linear betas 0.00085..0.012, 1000 steps) for the synthetic pipe:
    DDPM-style variance-preserving add_noise for '1-5' (PNDM family, identity scale_model_input) and the
    sigma parameterisation for 'xl' (EulerDiscrete: x + sigma*noise, x/sqrt(sigma^2+1))."""

    def __init__(self, euler):
        betas = torch.linspace(0.00085 ** 0.5, 0.012 ** 0.5, 1000, dtype=torch.float64) ** 2
        self.alphas_cumprod = torch.cumprod(1.0 - betas, 0).float()
        self.euler = euler
        self.timesteps = None

    def set_timesteps(self, n, device=None):
        self.timesteps = torch.arange(n - 1, -1, -1, device=device) * (1000 // n)

    def _sigma(self, t):
        ac = self.alphas_cumprod[int(t.flatten()[0])]
        return float(((1 - ac) / ac) ** 0.5)

    def scale_model_input(self, x, t):
        if not self.euler:
            return x
        return x / (self._sigma(t) ** 2 + 1) ** 0.5

    def add_noise(self, x, noise, t):
        ac = float(self.alphas_cumprod[int(t.flatten()[0])])
        if self.euler:
            return x + noise * self._sigma(t)
        return ac ** 0.5 * x + (1 - ac) ** 0.5 * noise


class SyntheticPipe:
    """Offline stand-in for the diffusers img2img pipeline object (`pipe`) used by FeatureExtractor."""

    def __init__(self, version, device, seed=0, stream_fp32=True):
        cfg = ARCH_CONFIGS[version]
        self.version = version
        self.device = device
        self.unet = NativeUNet(cfg, device=device, stream_fp32=stream_fp32).init_synthetic(seed)
        empty = types.SimpleNamespace(parameters=lambda: iter(()), to=lambda *a, **k: None)
        self.vae = types.SimpleNamespace(parameters=lambda: iter(()), config=types.SimpleNamespace(scaling_factor=0.13025))
        self.text_encoder = empty
        if cfg["addition_embed_text_time"]:
            pooled = cfg["add_in_dim"] - 6 * cfg["addition_time_embed_dim"]
            self.text_encoder_2 = types.SimpleNamespace(parameters=lambda: iter(()), to=lambda *a, **k: None,
                                                        config=types.SimpleNamespace(projection_dim=pooled))
        self.scheduler = _Scheduler(euler=bool(cfg["addition_embed_text_time"]))
        self.config = types.SimpleNamespace(requires_aesthetics_score=False)
        self.image_processor = types.SimpleNamespace(preprocess=self._preprocess)
        self._cfg = cfg

    # -- upstream stand-ins (deterministic, NOT the real encoders) -------------------------------
    def _preprocess(self, img):
        import numpy as np
        imgs = img if isinstance(img, (list, tuple)) else [img]
        out = []
        for im in imgs:
            if torch.is_tensor(im):
                out.append(im.float()[None] if im.dim() == 3 else im.float())
            else:
                a = torch.from_numpy(np.asarray(im, dtype=np.float32) / 255.0).permute(2, 0, 1)[None]
                out.append(a * 2.0 - 1.0)
        return torch.cat(out, 0)

    def _embeds(self, text, shape):
        seed = int.from_bytes(hashlib.sha256(text.encode()).digest()[:4], "little")
        g = torch.Generator().manual_seed(seed)
        return torch.randn(shape, generator=g).to(self.device, torch.float16)

    def encode_prompt(self, prompt, device=None, num_images_per_prompt=1, negative_prompt='',
                      do_classifier_free_guidance=True):
        cd = self._cfg["cross_attention_dim"]
        pe, ne = self._embeds(prompt, (1, 77, cd)), self._embeds("neg:" + (negative_prompt or ''), (1, 77, cd))
        if self._cfg["addition_embed_text_time"]:
            pd = self._cfg["add_in_dim"] - 6 * self._cfg["addition_time_embed_dim"]
            return pe, ne, self._embeds("pool:" + prompt, (1, pd)), self._embeds("npool:", (1, pd))
        return pe, ne

    def get_timesteps(self, num_inference_steps, strength, device):
        init = min(int(num_inference_steps * strength), num_inference_steps)
        t_start = max(num_inference_steps - init, 0)
        ts = self.scheduler.timesteps[t_start:]
        return ts, num_inference_steps - t_start

    def prepare_latents(self, image, timestep, batch_size, num_images_per_prompt, dtype, device, generator=None):
        """Synthetic 'VAE encode': 8x8 average pooling + a fixed 3->4 channel mix, then scheduler noise."""
        x = image.to(device, torch.float32)
        x = torch.nn.functional.avg_pool2d(x, 8)
        mix = torch.tensor([[0.6, 0.3, 0.1], [-0.2, 0.5, 0.4], [0.3, -0.4, 0.5], [0.2, 0.2, -0.6]], device=device)
        lat = torch.einsum("oc,bchw->bohw", mix, x) * 4.0
        g = torch.Generator(device=device).manual_seed(1234)
        noise = torch.randn(lat.shape, generator=g, device=device)
        return self.scheduler.add_noise(lat, noise, timestep).to(dtype)


def _native_from_diffusers(pipe, device):
    """Swap pipe.unet (diffusers UNet2DConditionModel) for the native implementation with the same weights."""
    unet = NativeUNet(config_from_diffusers(pipe.unet.config), device=device)
    unet.load_state_dict(pipe.unet.state_dict())
    pipe.unet = unet
    return pipe


def get_diffusion_model(version, dtype, offline_lora=None, offline_lora_filename=None, device="cuda"):
    dt = _parse_dtype(dtype)
    if version in _LATER:
        raise NotImplementedError(f"version '{version}' is not on the native hot path yet (SURVEY.md §8f / Appendix D)")
    if version not in _HF:
        raise NotImplementedError                                 # reference models.py:173-174
    if os.environ.get("GDF_SYNTHETIC_WEIGHTS", "0") not in ("", "0"):
        return SyntheticPipe(version, device, seed=int(os.environ.get("GDF_SYNTHETIC_SEED", "0")))
    try:
        import diffusers
    except ImportError as e:
        raise RuntimeError("diffusers is not installed and GDF_SYNTHETIC_WEIGHTS is not set: the text encoders / VAE / "
                           "checkpoint loading upstream of the native UNet come from diffusers (see INTEGRATION.md)") from e
    repo, cls = _HF[version]
    pipe = getattr(diffusers, cls).from_pretrained(repo, torch_dtype=dt, variant="fp16" if dt == torch.float16 else None)
    if version != "1-5":
        pipe.scheduler = diffusers.EulerDiscreteScheduler.from_config(pipe.scheduler.config)
    if offline_lora:
        pipe.load_lora_weights(offline_lora, weight_name=offline_lora_filename)
        pipe.fuse_lora()
    pipe = pipe.to(device)
    return _native_from_diffusers(pipe, device)
