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

from .native import (ARCH_CONFIGS, FLUX_CONFIGS, PIXART_CONFIGS, VAE_CONFIGS, NativeFluxTransformer, NativePixArtTransformer,
                     NativeUNet, NativeVAEDecoder, NativeVAEEncoder, config_from_diffusers)

# version -> (HF repo id, pipeline class name) exactly as the reference selects them (models.py:18-70)
_HF = {
    "1-5": ("stable-diffusion-v1-5/stable-diffusion-v1-5", "StableDiffusionImg2ImgPipeline"),
    "2-1": ("stabilityai/stable-diffusion-2-1-base", "StableDiffusionImg2ImgPipeline"),      # + EulerDiscreteScheduler (:38-39)
    "xl": ("stabilityai/stable-diffusion-xl-base-1.0", "StableDiffusionXLImg2ImgPipeline"),
    "pgv2": ("playgroundai/playground-v2-1024px-aesthetic", "StableDiffusionXLImg2ImgPipeline"),
}
_LATER = ("if", "hunyuan")                       # no BASELINE config (pixel-space UNet / untested DiT)


def _fill(model, loader):
    """Weights of a native model.  Single process — or a process group the launch did not opt in with (components/dist.py
    enable_weight_broadcast: extract_feature.py / bench.py under torchrun do) —: just `loader(model)` (checkpoint re-layout or
    synthetic init).  Data-parallel launch: rank 0 runs the loader, every other rank receives the flat device arena over RCCL
    (broadcast_model_weights; the arena sizes are checked to agree first)."""
    from . import dist as D
    if not D.weight_broadcast_enabled():
        loader(model)
        return model
    rank, _world = D.rank_world()
    if rank == 0:
        loader(model)
    return D.broadcast_model_weights(model)


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
        a, b = self.noise_scalars(t)
        return a * x + b * noise

    def step_scalars(self, t):
        """(c_sample, c_eps) with step(eps, t, x).prev_sample = c_sample x + c_eps eps for the FIRST step after set_timesteps:
        PNDM's first step_plms call (eq. 9 of the PNDM paper with the model output as is) for '1-5', the Euler step
        x + (sigma_next - sigma) eps otherwise (epsilon prediction; un-vendored diffusers schedulers, oracle/vae_ref.py)."""
        ti = int(t.flatten()[0])
        if self.euler:
            idx = int((self.timesteps == ti).nonzero()[0]) if self.timesteps is not None else None
            nxt = int(self.timesteps[idx + 1]) if idx is not None and idx + 1 < len(self.timesteps) else -1
            acn = float(self.alphas_cumprod[nxt]) if nxt >= 0 else 1.0
            sig_next = ((1 - acn) / acn) ** 0.5 if nxt >= 0 else 0.0
            return 1.0, sig_next - self._sigma(t)
        step = 1000 // len(self.timesteps) if self.timesteps is not None else 1
        a_t = float(self.alphas_cumprod[ti])
        a_p = float(self.alphas_cumprod[ti - step]) if ti - step >= 0 else float(self.alphas_cumprod[0])
        den = a_t * (1 - a_p) ** 0.5 + (a_t * (1 - a_t) * a_p) ** 0.5
        return (a_p / a_t) ** 0.5, -(a_p - a_t) / den

    def step(self, model_output, t, sample, return_dict=False):
        a, b = self.step_scalars(t)
        return (a * sample + b * model_output,)

    def noise_scalars(self, t):
        """(noise_a, noise_b) with add_noise(x, n, t) = a x + b n."""
        ac = float(self.alphas_cumprod[int(t.flatten()[0])])
        if self.euler:
            return 1.0, self._sigma(t)
        return ac ** 0.5, (1 - ac) ** 0.5


def scheduler_noise_scalars(scheduler, timestep):
    """(a, b) such that scheduler.add_noise(x, noise, timestep) == a * x + b * noise for the call `prepare_latents` of the img2img pipelines makes
    (reference feature/diffusion_feature.py:371-380 -> diffusers prepare_latents -> scheduler.add_noise).  Schedulers that say so themselves
    (`noise_scalars`: the synthetic pipe's) are asked; a diffusers scheduler is PROBED on a deep copy with one-element CPU tensors — (x, noise) =
    (1, 0) and (0, 1), linearity verified on a third point — so every family the reference configures gives its own coefficients: PNDM / DDPM
    (sqrt(ac), sqrt(1 - ac); '1-5'), EulerDiscrete (1, sigma; '2-1', 'xl', 'pgv2': models.py:26,38,51), DPMSolverMultistep
    (alpha_t, sigma_t; the PixArt pipelines).  (Until round 6 this read `scheduler.sigmas` directly, which is the Euler rule — wrong for
    DPMSolverMultistep, which also has `sigmas` and `index_for_timestep`.)"""
    if hasattr(scheduler, "noise_scalars"):
        return scheduler.noise_scalars(timestep)
    import copy
    t1 = timestep.flatten()[:1] if torch.is_tensor(timestep) else torch.as_tensor([timestep])

    def probe(x, n):
        sch = copy.deepcopy(scheduler)
        one = lambda v: torch.full((1, 1, 1, 1), float(v), dtype=torch.float64)
        return float(sch.add_noise(one(x), one(n), t1.cpu()).flatten()[0])
    a, b = probe(1.0, 0.0), probe(0.0, 1.0)
    chk = probe(0.5, -2.0)
    if abs(chk - (0.5 * a - 2.0 * b)) > 1e-6 * (1.0 + abs(chk)):
        raise NotImplementedError("scheduler.add_noise is not linear in (sample, noise) for this scheduler configuration")
    return a, b


def scheduler_step_scalars(scheduler, timestep):
    """(c_sample, c_eps) such that `scheduler.step(noise_pred, timestep, latents)[0] == c_sample * latents + c_eps * noise_pred` for the
    call the reference makes at feature/diffusion_feature.py:478-480 (the first step after set_timesteps).  Schedulers that say so
    themselves (`step_scalars`) are asked; for a diffusers scheduler the two coefficients are PROBED on a deep copy with two
    one-element CPU tensors — (x, eps) = (1, 0) and (0, 1) — because the first PNDM / Euler step is linear in both (epsilon / v
    prediction without thresholding or clipping), which the probe verifies on a third point."""
    if hasattr(scheduler, "step_scalars"):
        return scheduler.step_scalars(timestep)
    import copy
    t0 = timestep.flatten()[0] if torch.is_tensor(timestep) else timestep

    def probe_on(dev, x, e):
        sch = copy.deepcopy(scheduler)
        t = t0.to(dev) if torch.is_tensor(t0) else t0
        out = sch.step(torch.full((1, 1, 1, 1), float(e), dtype=torch.float64, device=dev), t,
                       torch.full((1, 1, 1, 1), float(x), dtype=torch.float64, device=dev), return_dict=False)[0]
        return float(out.flatten()[0])

    def probe(x, e):
        try:
            return probe_on("cpu", x, e)                 # diffusers keeps sigmas / alphas_cumprod on the CPU
        except (RuntimeError, TypeError):
            if torch.is_tensor(t0) and t0.device.type != "cpu":
                return probe_on(t0.device, x, e)         # a scheduler whose tables live on the timestep's device
            raise
    a, b = probe(1.0, 0.0), probe(0.0, 1.0)
    chk = probe(0.5, -2.0)
    if abs(chk - (0.5 * a - 2.0 * b)) > 1e-6 * (1.0 + abs(chk)):
        raise NotImplementedError("scheduler.step is not linear in (sample, model_output) for this scheduler configuration; "
                                  "'vae-out' needs a linear first step (PNDM / EulerDiscrete as the reference configures them)")
    return a, b


def native_vae_decoder(pipe, device):
    """The pipe's native AutoencoderKL decoder, created on first use (only a config that asks for 'vae-out' needs it)."""
    dec = getattr(pipe, "_native_vae_decoder", None)
    if dec is None:
        if getattr(pipe, "native_vae", None) is not None and not hasattr(pipe.vae, "state_dict"):      # synthetic pipe
            dec = _fill(NativeVAEDecoder(VAE_CONFIGS["sd"], device=device), lambda m: m.init_synthetic(getattr(pipe, "_seed", 0) + 2))
        else:
            vc = pipe.vae.config
            dec = NativeVAEDecoder(dict(in_channels=vc.in_channels, latent_channels=vc.latent_channels,
                                        block_out_channels=tuple(vc.block_out_channels), layers_per_block=vc.layers_per_block,
                                        use_quant_conv=int(getattr(vc, "use_post_quant_conv", getattr(vc, "use_quant_conv", True)))),
                                   device=device)
            _fill(dec, lambda m: m.load_vae_state_dict(pipe.vae.state_dict()))
        pipe._native_vae_decoder = dec
    return dec


def native_prepare_latents(pipe, image, timestep, batch_size, num_images_per_prompt, dtype, device, generator=None):
    """Drop-in for `StableDiffusion(XL)Img2ImgPipeline.prepare_latents` as the reference calls it
    (feature/diffusion_feature.py:371-380): VAE encode -> latent_dist.sample -> * scaling_factor -> scheduler.add_noise,
    executed by libgdf.so (include/gdf_vae.h); the random tensors come from torch exactly where the pipeline draws them."""
    enc = pipe.native_vae
    f = 1 << (len(enc.cfg["block_out_channels"]) - 1)
    B, _, H, W = image.shape
    shape = (B, enc.cfg["latent_channels"], H // f, W // f)
    eps = torch.randn(shape, generator=generator, device=device, dtype=torch.float32)
    noise = torch.randn(shape, generator=generator, device=device, dtype=torch.float32)
    a, b = scheduler_noise_scalars(pipe.scheduler, timestep)
    lat = enc.encode(image, eps=eps, noise=noise, scaling_factor=float(pipe.vae.config.scaling_factor), noise_a=a, noise_b=b)
    return lat.to(dtype)


class SyntheticPipe:
    """Offline stand-in for the diffusers img2img pipeline object (`pipe`) used by FeatureExtractor."""
    synthetic_weights = True                      # seeded N(0, 1/fan_in) weights: the statistics the operand-plan table was made on

    def __init__(self, version, device, seed=0, stream_fp32=True):
        cfg = ARCH_CONFIGS[version]
        self.version = version
        self.device = device
        self._seed = seed
        self.unet = _fill(NativeUNet(cfg, device=device, stream_fp32=stream_fp32), lambda m: m.init_synthetic(seed))
        empty = types.SimpleNamespace(parameters=lambda: iter(()), to=lambda *a, **k: None)
        self.vae = types.SimpleNamespace(parameters=lambda: iter(()), config=types.SimpleNamespace(
            scaling_factor=0.13025 if cfg["addition_embed_text_time"] else 0.18215))
        # true-architecture AutoencoderKL encoder (seeded random weights) in libgdf.so: the step before the hot path
        self.native_vae = _fill(NativeVAEEncoder(VAE_CONFIGS["sd"], device=device), lambda m: m.init_synthetic(seed + 1))
        self.text_encoder = empty
        if cfg["addition_embed_text_time"]:
            pooled = cfg["add_in_dim"] - 6 * cfg["addition_time_embed_dim"]
            self.text_encoder_2 = types.SimpleNamespace(parameters=lambda: iter(()), to=lambda *a, **k: None,
                                                        config=types.SimpleNamespace(projection_dim=pooled))
        self.scheduler = _Scheduler(euler=bool(cfg["addition_embed_text_time"]) or version == "2-1")   # reference models.py:26,38,51
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
        """prepare_latents of the img2img pipelines on the native VAE encoder (random-weight AutoencoderKL offline)."""
        g = generator or torch.Generator(device=device).manual_seed(1234)
        return native_prepare_latents(self, image.to(device), timestep, batch_size, num_images_per_prompt, dtype, device, g)


class SyntheticPixartPipe(SyntheticPipe):
    """Offline stand-in for diffusers' PixArtSigmaPipeline as FeatureExtractor uses it: true-architecture DiT
    (NativePixArtTransformer, seeded random weights), native VAE encoder, stand-in T5 embeddings with a ragged mask."""

    def __init__(self, version, device, seed=0, cfg=None, n_txt=None):
        cfg = dict(cfg or PIXART_CONFIGS[version])
        alpha = version == "pixart-alpha"                         # PixArtAlphaPipeline: max_sequence_length 120, sd-vae-ft-ema
        n_txt = n_txt or (120 if alpha else 300)
        self.version = version
        self.device = device
        self._pcfg = cfg
        self.n_txt = n_txt
        self.transformer = _fill(NativePixArtTransformer(cfg, device=device), lambda m: m.init_synthetic(seed))
        self.unet = self.transformer               # reference models.py:91 `pipe.unet = pipe.transformer`
        empty = types.SimpleNamespace(parameters=lambda: iter(()), to=lambda *a, **k: None)
        self.vae = types.SimpleNamespace(parameters=lambda: iter(()), config=types.SimpleNamespace(scaling_factor=0.18215 if alpha else 0.13025))
        self.native_vae = _fill(NativeVAEEncoder(VAE_CONFIGS["sd"], device=device), lambda m: m.init_synthetic(seed + 1))
        self.text_encoder = empty
        self.scheduler = _Scheduler(euler=False)
        self.config = types.SimpleNamespace(requires_aesthetics_score=False)
        self.image_processor = types.SimpleNamespace(preprocess=self._preprocess)

    def encode_prompt(self, prompt, device=None, num_images_per_prompt=1, negative_prompt='',
                      do_classifier_free_guidance=True, **kw):
        """(prompt_embeds, prompt_attention_mask, negative_prompt_embeds, negative_prompt_attention_mask) — PixArt order."""
        cc = self._pcfg["caption_channels"]
        n_valid = max(1, min(self.n_txt, len(prompt.split()) + 2))
        mask = (torch.arange(self.n_txt, device=self.device)[None] < n_valid).to(torch.int64)
        return self._embeds(prompt, (1, self.n_txt, cc)), mask, self._embeds("neg:", (1, self.n_txt, cc)), torch.ones_like(mask)


class SyntheticFluxPipe:
    """Offline stand-in for the reference's PATCHED FluxImg2ImgPipeline as `FeatureExtractor.extract` drives it
    (`pipe(image=..., prompt=..., strength=t/1000, guidance_scale=1)`, diffusion_feature.py:246-254): true-architecture
    MMDiT (NativeFluxTransformer, seeded random weights) + deterministic stand-ins for the T5/CLIP encoders and the
    16-channel VAE, flow-matching sigmas with the resolution-dependent shift, 2x2 latent packing and the img2img
    strength -> timestep rule.  Like the reference's pipeline (feature/diffusers/pipelines/flux/pipeline_flux_img2img.py:
    804-841: `return` at the end of the FIRST loop iteration) one call runs EXACTLY ONE transformer forward, at
    sigmas[t_start], and returns None."""
    synthetic_weights = True

    num_inference_steps = 28                      # FluxImg2ImgPipeline.__call__ default
    returns_after_first_forward = True            # like the reference's patched pipeline (:841)
    # FlowMatchEulerDiscreteScheduler config of black-forest-labs/FLUX.1-dev (scheduler/scheduler_config.json) [memory]
    sched_cfg = dict(base_image_seq_len=256, max_image_seq_len=4096, base_shift=0.5, max_shift=1.15)

    def __init__(self, device, seed=0, cfg=None, n_txt=512):
        self.device = device
        self._cfg = dict(cfg or FLUX_CONFIGS["flux"])
        self.n_txt = n_txt
        self.transformer = _fill(NativeFluxTransformer(self._cfg, device=device, compute_dtype=flux_compute_dtype()), lambda m: m.init_synthetic(seed))
        self.unet = self.transformer               # reference models.py:169 `pipe.unet = pipe.transformer`
        empty = types.SimpleNamespace(parameters=lambda: iter(()), to=lambda *a, **k: None)
        self.vae = types.SimpleNamespace(parameters=lambda: iter(()), config=types.SimpleNamespace(scaling_factor=0.3611, shift_factor=0.1159))
        self.text_encoder = empty
        self.text_encoder_2 = empty
        self.scheduler = types.SimpleNamespace()
        self.image_processor = types.SimpleNamespace(preprocess=SyntheticPipe._preprocess.__get__(self))
        self.last_call = None                      # (sigma, packed latents, prompt embeds, ...) of the last call: test / debugging aid

    def _embeds(self, text, shape):
        seed = int.from_bytes(hashlib.sha256(text.encode()).digest()[:4], "little")
        return torch.randn(shape, generator=torch.Generator().manual_seed(seed)).to(self.device, torch.float16)

    def sigmas(self, n_steps, image_seq_len):
        """pipeline_flux_img2img.py:744-760: sigmas = linspace(1, 1/N, N), shifted by mu = calculate_shift(image_seq_len, ...)
        (`time_shift`: exp(mu) / (exp(mu) + (1/sigma - 1)), FlowMatchEulerDiscreteScheduler with dynamic shifting), + final 0."""
        c = self.sched_cfg
        m = (c["max_shift"] - c["base_shift"]) / (c["max_image_seq_len"] - c["base_image_seq_len"])
        mu = image_seq_len * m + (c["base_shift"] - m * c["base_image_seq_len"])
        s = torch.linspace(1.0, 1.0 / n_steps, n_steps, dtype=torch.float64)
        s = math.exp(mu) / (math.exp(mu) + (1.0 / s - 1.0))
        return s.tolist() + [0.0]

    def __call__(self, image=None, prompt=None, strength=0.6, guidance_scale=7.0, num_inference_steps=None, **kw):
        dev = self.device
        imgs = image if isinstance(image, (list, tuple)) else [image]
        x = torch.cat([self.image_processor.preprocess(i) for i in imgs], 0).to(dev, torch.float32)
        B = x.shape[0]
        prompts = prompt if isinstance(prompt, (list, tuple)) else [prompt] * B
        enc = torch.cat([self._embeds(p, (1, self.n_txt, self._cfg["joint_attention_dim"])) for p in prompts], 0)
        pooled = torch.cat([self._embeds("pool:" + p, (1, self._cfg["pooled_projection_dim"])) for p in prompts], 0)
        # synthetic 16-channel 'VAE encode' (8x8 average pooling + fixed channel mix), then 2x2 packing -> (B, S, 64)
        lat = torch.nn.functional.avg_pool2d(x, 8)
        mix = torch.linspace(-1.0, 1.0, 48, device=dev).reshape(16, 3)
        lat = torch.einsum("oc,bchw->bohw", mix, lat) * 2.0
        Bc, Cc, H, W = lat.shape
        gh, gw = H // 2, W // 2
        pack = lambda z: z.view(Bc, Cc, gh, 2, gw, 2).permute(0, 2, 4, 1, 3, 5).reshape(Bc, gh * gw, Cc * 4)
        # img2img schedule (get_timesteps, :606-616): the last int(N * strength) steps remain; only the FIRST of them is run
        N = num_inference_steps or self.num_inference_steps
        sigmas = self.sigmas(N, gh * gw)
        init = min(N * strength, N)
        t_start = int(max(N - init, 0))
        if N - t_start < 1:
            raise ValueError(f"After adjusting the num_inference_steps by strength parameter: {strength}, the number of "
                             f"pipeline steps is {N - t_start} which is < 1 and not appropriate for this pipeline.")
        noise = torch.randn(lat.shape, generator=torch.Generator(device=dev).manual_seed(1234), device=dev)
        s0 = sigmas[t_start]
        z = pack((1.0 - s0) * lat + s0 * noise)                      # FlowMatchEulerDiscreteScheduler.scale_noise
        img_ids = torch.zeros(gh, gw, 3, device=dev)
        img_ids[..., 1] = torch.arange(gh, device=dev)[:, None]; img_ids[..., 2] = torch.arange(gw, device=dev)[None, :]
        img_ids = img_ids.reshape(gh * gw, 3)
        txt_ids = torch.zeros(self.n_txt, 3, device=dev)
        guidance = torch.full((B,), float(guidance_scale), device=dev) if self._cfg["guidance_embeds"] else None
        t = torch.full((B,), s0, device=dev)                          # the pipeline passes timestep / 1000 = sigma (:816)
        self.last_call = dict(sigma=s0, t_start=t_start, hidden_states=z, encoder_hidden_states=enc, pooled_projections=pooled,
                              img_ids=img_ids, txt_ids=txt_ids, guidance=guidance, grid=(gh, gw))
        # ONE denoiser forward through the model's __call__ (it delivers the hooks to the FeatureStore), then return like the
        # reference's patched loop does (:841) — no scheduler.step, no further steps, no VAE decode
        self.transformer(hidden_states=z, timestep=t, guidance=guidance, pooled_projections=pooled, encoder_hidden_states=enc,
                         txt_ids=txt_ids, img_ids=img_ids, joint_attention_kwargs=None, return_dict=False, grid=(gh, gw))
        return None


def flux_compute_dtype():
    """Arithmetic of the Flux transformer behind FeatureExtractor: 'auto' (round 5 default: fp16 operands with range scaling, every hook within
    1e-3 of the fp32 reference at the bf16 mode's speed; NativeFluxTransformer) unless GDF_FLUX_DTYPE names another mode ('bfloat16' = the
    reference's own dtype, 'bfloat16x2', 'float16', 'fp8-mx')."""
    return os.environ.get("GDF_FLUX_DTYPE", "") or "auto"


def flux_config_from_diffusers(c):
    """FluxTransformer2DModel `.config` -> the fields of FLUX_CONFIGS"""
    return dict(in_channels=c.in_channels, num_layers=c.num_layers, num_single_layers=c.num_single_layers,
                attention_head_dim=c.attention_head_dim, num_attention_heads=c.num_attention_heads,
                joint_attention_dim=c.joint_attention_dim, pooled_projection_dim=c.pooled_projection_dim,
                guidance_embeds=int(bool(c.guidance_embeds)), axes_dims_rope=tuple(c.axes_dims_rope), mlp_ratio=4)


def _native_flux_from_diffusers(pipe, device):
    """Swap pipe.transformer (diffusers FluxTransformer2DModel, bf16) for the native MMDiT with the same weights."""
    cfg = flux_config_from_diffusers(pipe.transformer.config)
    net = NativeFluxTransformer(cfg, device=device, compute_dtype=flux_compute_dtype())
    sd = pipe.transformer.state_dict()
    _fill(net, lambda m: m.load_state_dict(sd))
    # the fp16 modes ('auto' -> 'float16s') were validated on seeded synthetic weights; the load-time guard only looks at WEIGHT
    # cast error.  A real checkpoint additionally gets an ACTIVATION range check on its first forward (NativeFluxTransformer.arm_range_check):
    # a saturated / non-finite 16-bit tensor anywhere in the model re-loads these weights in 'bfloat16x2' (bf16's range, the reference's dtype
    # as operand pairs) with one warning.  The state dict is held until that forward has run.
    net.arm_range_check(sd)
    pipe.transformer = net
    pipe.unet = net
    return pipe


def pixart_config_from_diffusers(c):
    """PixArtTransformer2DModel / Transformer2DModel `.config` -> the fields of PIXART_CONFIGS (interpolation_scale None = diffusers' default
    max(sample_size // 64, 1))"""
    ss = int(c.sample_size)
    isc = getattr(c, "interpolation_scale", None)
    if getattr(c, "use_additional_conditions", None):
        raise NotImplementedError("PixArt-alpha micro-conditioning (use_additional_conditions) is not native")
    return dict(num_attention_heads=int(c.num_attention_heads), attention_head_dim=int(c.attention_head_dim), in_channels=int(c.in_channels),
                out_channels=int(c.out_channels), num_layers=int(c.num_layers), patch_size=int(c.patch_size), sample_size=ss,
                caption_channels=int(c.caption_channels), interpolation_scale=int(isc) if isc is not None else max(ss // 64, 1))


def _img2img_get_timesteps(self, num_inference_steps, strength, device, denoising_start=None):
    """`get_timesteps` of the img2img pipelines, for pipelines that do not have one: the STOCK PixArt pipelines are text-to-image (the reference
    carries patched copies that add this method and an image-taking prepare_latents: feature/diffusers/pipelines/pixart_alpha/
    pipeline_pixart_sigma.py:598-700).  Same rule: the last int(N * strength) steps remain."""
    init_timestep = min(int(num_inference_steps * strength), num_inference_steps)
    t_start = max(num_inference_steps - init_timestep, 0)
    order = getattr(self.scheduler, "order", 1)
    timesteps = self.scheduler.timesteps[t_start * order:]
    if hasattr(self.scheduler, "set_begin_index"):
        self.scheduler.set_begin_index(t_start * order)
    return timesteps, num_inference_steps - t_start


def _native_vae_from_diffusers(pipe, device):
    """pipe.native_vae = the AutoencoderKL encoder half in libgdf.so with pipe.vae's weights; pipe.prepare_latents = native_prepare_latents"""
    vc = pipe.vae.config
    enc = NativeVAEEncoder(dict(in_channels=vc.in_channels, latent_channels=vc.latent_channels,
                                block_out_channels=tuple(vc.block_out_channels), layers_per_block=vc.layers_per_block,
                                use_quant_conv=int(getattr(vc, "use_quant_conv", True))), device=device)
    _fill(enc, lambda m: m.load_vae_state_dict(pipe.vae.state_dict()))
    pipe.native_vae = enc
    # (bound to a weak proxy: `pipe.prepare_latents = MethodType(f, pipe)` would make the pipeline a reference cycle, and a cycle is freed by the
    #  cyclic collector at an arbitrary later allocation — in whatever thread that happens to run — together with its multi-GB device arenas)
    import weakref
    pipe.prepare_latents = types.MethodType(native_prepare_latents, weakref.proxy(pipe))
    return pipe


def _native_from_diffusers(pipe, device):
    """Swap pipe.unet (diffusers UNet2DConditionModel) for the native implementation with the same weights."""
    cfg = getattr(pipe, "_gdf_unet_config", None) or config_from_diffusers(pipe.unet.config)   # ranks > 0 may have no UNet module
    unet = NativeUNet(cfg, device=device)
    _fill(unet, lambda m: m.load_state_dict(pipe.unet.state_dict()))
    pipe.unet = unet
    # the step before the hot path: VAE encode + sample + noise-add in libgdf.so as well.
    # GDF_NATIVE_VAE=0 keeps diffusers' prepare_latents (e.g. the original SDXL VAE, whose activations need fp32).
    if os.environ.get("GDF_NATIVE_VAE", "1") not in ("", "0"):
        _native_vae_from_diffusers(pipe, device)
    return pipe


def get_diffusion_model(version, dtype, offline_lora=None, offline_lora_filename=None, device="cuda"):
    dt = _parse_dtype(dtype)
    if version in _LATER:
        raise NotImplementedError(f"version '{version}' is not on the native hot path yet (SURVEY.md §8f / Appendix D)")
    synthetic = os.environ.get("GDF_SYNTHETIC_WEIGHTS", "0") not in ("", "0")
    if version == "flux":                                         # reference models.py:150-170 (bf16 pipeline, fp16 hooks)
        if synthetic:
            return SyntheticFluxPipe(device, seed=int(os.environ.get("GDF_SYNTHETIC_SEED", "0")))
        try:
            import diffusers
        except ImportError as e:
            raise RuntimeError("diffusers is not installed and GDF_SYNTHETIC_WEIGHTS is not set (see INTEGRATION.md)") from e
        pipe = diffusers.FluxImg2ImgPipeline.from_pretrained('black-forest-labs/FLUX.1-dev', torch_dtype=torch.bfloat16,
                                                             use_safetensors=True)
        if offline_lora:
            pipe.load_lora_weights(offline_lora, weight_name=offline_lora_filename)
            pipe.fuse_lora()
        return _native_flux_from_diffusers(pipe.to(device), device)
    if version in PIXART_CONFIGS:                                 # reference models.py:72-111 (PixArtSigmaPipeline)
        if synthetic:
            return SyntheticPixartPipe(version, device, seed=int(os.environ.get("GDF_SYNTHETIC_SEED", "0")))
        try:
            import diffusers
        except ImportError as e:
            raise RuntimeError("diffusers is not installed and GDF_SYNTHETIC_WEIGHTS is not set (see INTEGRATION.md)") from e
        if version == "pixart-alpha":                             # reference models.py:103-115 (120-token T5 captions, SD VAE)
            pipe = diffusers.PixArtAlphaPipeline.from_pretrained("PixArt-alpha/PixArt-XL-2-512x512", torch_dtype=dt, variant="fp16",
                                                                 use_safetensors=True).to(device)
        else:
            repo = "PixArt-alpha/PixArt-Sigma-XL-2-1024-MS" if version == "pixart-sigma" else "PixArt-alpha/PixArt-Sigma-XL-2-512-MS"
            pipe = diffusers.PixArtSigmaPipeline.from_pretrained(repo, torch_dtype=dt, use_safetensors=True).to(device)
        # the architecture comes from the LOADED module (PIXART_CONFIGS holds the same numbers for the synthetic pipes)
        net = NativePixArtTransformer(pixart_config_from_diffusers(pipe.transformer.config), device=device)
        _fill(net, lambda m: m.load_state_dict({k: v for k, v in pipe.transformer.state_dict().items() if k != "pos_embed.pos_embed"}))
        pipe.transformer = pipe.unet = net
        # a STOCK diffusers PixArt pipeline is text-to-image: no `get_timesteps`, no image-taking `prepare_latents` (the reference patches both
        # in).  Round 6: the img2img front half is supplied here — the native VAE encoder + scheduler.add_noise for prepare_latents (unless
        # GDF_NATIVE_VAE=0 and the installed pipeline is the reference's patched one), the img2img rule for get_timesteps
        if os.environ.get("GDF_NATIVE_VAE", "1") not in ("", "0"):
            _native_vae_from_diffusers(pipe, device)
        if not hasattr(pipe, "get_timesteps"):
            import weakref
            pipe.get_timesteps = types.MethodType(_img2img_get_timesteps, weakref.proxy(pipe))
        return pipe
    if version not in _HF:
        raise NotImplementedError                                 # reference models.py:173-174
    if synthetic:
        return SyntheticPipe(version, device, seed=int(os.environ.get("GDF_SYNTHETIC_SEED", "0")))
    try:
        import diffusers
    except ImportError as e:
        raise RuntimeError("diffusers is not installed and GDF_SYNTHETIC_WEIGHTS is not set: the text encoders / VAE / "
                           "checkpoint loading upstream of the native UNet come from diffusers (see INTEGRATION.md)") from e
    repo, cls = _HF[version]
    kw = dict(variant="fp16") if (version in ("xl", "pgv2") and dt == torch.float16) else {}      # reference models.py:51-53
    from . import dist as D
    rank, world = D.rank_world()
    # data-parallel launch (opt-in, see _fill): only rank 0 reads the 5 GB UNet checkpoint, the other ranks receive the re-laid-out
    # arena.  With an offline LoRA every rank loads its own UNet instead: load_lora_weights / fuse_lora need the module, and each
    # rank fuses the same weights (the broadcast then only overwrites them with rank 0's identical arena).
    skip_unet = D.weight_broadcast_enabled() and rank != 0 and not offline_lora
    if skip_unet:
        kw["unet"] = None
    pipe = getattr(diffusers, cls).from_pretrained(repo, torch_dtype=dt, use_safetensors=True, **kw)
    if version != "1-5":
        pipe.scheduler = diffusers.EulerDiscreteScheduler.from_config(pipe.scheduler.config)
    if offline_lora:
        pipe.load_lora_weights(offline_lora, weight_name=offline_lora_filename)
        pipe.fuse_lora()
    # the architecture descriptor comes from rank 0's LOADED module (constructor defaults filled in: the raw config.json of SD1.5 /
    # SD2.1 lacks `transformer_layers_per_block`, SD1.5's also `use_linear_projection`) and is sent to the ranks without a UNet
    cfg = config_from_diffusers(pipe.unet.config) if pipe.unet is not None else None
    if D.weight_broadcast_enabled():
        cfg = D.broadcast_object(cfg, src=0)
    pipe._gdf_unet_config = cfg
    pipe = pipe.to(device)
    return _native_from_diffusers(pipe, device)
