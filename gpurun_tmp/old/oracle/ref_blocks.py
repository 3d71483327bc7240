"""ORACLE SUPPORT (test infrastructure, build container only).

Imports the reference's own patched-diffusers block modules *from where they lie*
under /root/reference (never copied) so that oracle/unet_ref.py can be validated
against them and golden vectors generated (tests/golden/gen_golden.py).

The reference files are fragments of the `diffusers==0.32.2` package (README.md:57,70) and use
relative imports into parts of diffusers that are NOT in /root/reference and are not
installed here.  Those parts are provided as minimal scaffolding modules below
(utility no-ops, config mixins, name-only placeholder classes) plus a restatement of
`activations.GEGLU/get_activation` from the published diffusers algorithm.  Nothing here
runs on the GPU box (no /root/reference there).
"""
import importlib.util
import inspect
import logging as _pylogging
import os
import sys
import types

import torch
import torch.nn.functional as F
from torch import nn

REF_ROOT = os.environ.get("GDF_REFERENCE_ROOT", "/root/reference")
_FEATURE = os.path.join(REF_ROOT, "feature")


def available():
    return os.path.isdir(os.path.join(_FEATURE, "diffusers", "models"))


def _mod(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


def _pkg(name, **attrs):
    m = _mod(name, **attrs)
    m.__path__ = []
    return m


class _Cfg(dict):
    __getattr__ = dict.__getitem__


def _register_to_config(init):
    sig = inspect.signature(init)

    def wrapped2(self, *a, **kw):
        ba = sig.bind(self, *a, **kw)
        ba.apply_defaults()
        self.__dict__["config"] = _Cfg({k: v for k, v in ba.arguments.items() if k != "self"})
        init(self, *a, **kw)
    return wrapped2


def _placeholder(name):
    return type(name, (nn.Module,), {"__init__": lambda self, *a, **k: (_ for _ in ()).throw(
        NotImplementedError(f"{name}: un-vendored diffusers class, not needed on the UNet hot path"))})


class _GEGLU(nn.Module):
    """diffusers==0.32.2 activations.GEGLU (un-vendored): proj -> chunk(2) -> hidden * gelu(gate)."""

    def __init__(self, dim_in, dim_out, bias=True):
        super().__init__()
        self.proj = nn.Linear(dim_in, dim_out * 2, bias=bias)

    def forward(self, hidden_states, *args, **kwargs):
        hidden_states, gate = self.proj(hidden_states).chunk(2, dim=-1)
        return hidden_states * F.gelu(gate)


def _get_activation(name):
    return {"swish": nn.SiLU(), "silu": nn.SiLU(), "mish": nn.Mish(), "gelu": nn.GELU(), "relu": nn.ReLU()}[name.lower()]



# ---- un-vendored diffusers==0.32.2 classes the Flux files import (transformer_flux.py:34-38,
# attention_processor.py:141,2331, attention.py:22) — restated from the published algorithm, parameter names as in
# the diffusers state_dict ---------------------------------------------------------------------------
class _GELU(nn.Module):
    """activations.GELU: proj -> F.gelu(approximate)."""

    def __init__(self, dim_in, dim_out, approximate="none", bias=True):
        super().__init__()
        self.proj = nn.Linear(dim_in, dim_out, bias=bias)
        self.approximate = approximate

    def forward(self, hidden_states):
        return F.gelu(self.proj(hidden_states), approximate=self.approximate)


class _RMSNorm(nn.Module):
    """normalization.RMSNorm (weight only)."""

    def __init__(self, dim, eps, elementwise_affine=True, bias=False):
        super().__init__()
        self.eps = eps
        self.weight = nn.Parameter(torch.ones(dim)) if elementwise_affine else None

    def forward(self, hidden_states):
        variance = hidden_states.to(torch.float32).pow(2).mean(-1, keepdim=True)
        hidden_states = hidden_states * torch.rsqrt(variance + self.eps)
        if self.weight is not None:
            hidden_states = hidden_states * self.weight
        return hidden_states


class _AdaLayerNormZero(nn.Module):
    def __init__(self, embedding_dim, num_embeddings=None, norm_type="layer_norm", bias=True):
        super().__init__()
        self.silu = nn.SiLU()
        self.linear = nn.Linear(embedding_dim, 6 * embedding_dim, bias=bias)
        self.norm = nn.LayerNorm(embedding_dim, elementwise_affine=False, eps=1e-6)

    def forward(self, x, timestep=None, class_labels=None, hidden_dtype=None, emb=None):
        emb = self.linear(self.silu(emb))
        shift_msa, scale_msa, gate_msa, shift_mlp, scale_mlp, gate_mlp = emb.chunk(6, dim=1)
        x = self.norm(x) * (1 + scale_msa[:, None]) + shift_msa[:, None]
        return x, gate_msa, shift_mlp, scale_mlp, gate_mlp


class _AdaLayerNormZeroSingle(nn.Module):
    def __init__(self, embedding_dim, norm_type="layer_norm", bias=True):
        super().__init__()
        self.silu = nn.SiLU()
        self.linear = nn.Linear(embedding_dim, 3 * embedding_dim, bias=bias)
        self.norm = nn.LayerNorm(embedding_dim, elementwise_affine=False, eps=1e-6)

    def forward(self, x, emb=None):
        emb = self.linear(self.silu(emb))
        shift_msa, scale_msa, gate_msa = emb.chunk(3, dim=1)
        x = self.norm(x) * (1 + scale_msa[:, None]) + shift_msa[:, None]
        return x, gate_msa


class _AdaLayerNormContinuous(nn.Module):
    def __init__(self, embedding_dim, conditioning_embedding_dim, elementwise_affine=True, eps=1e-5, bias=True,
                 norm_type="layer_norm"):
        super().__init__()
        self.silu = nn.SiLU()
        self.linear = nn.Linear(conditioning_embedding_dim, embedding_dim * 2, bias=bias)
        self.norm = nn.LayerNorm(embedding_dim, eps, elementwise_affine, bias)

    def forward(self, x, conditioning_embedding):
        emb = self.linear(self.silu(conditioning_embedding).to(x.dtype))
        scale, shift = torch.chunk(emb, 2, dim=1)
        return self.norm(x) * (1 + scale)[:, None, :] + shift[:, None, :]


def _get_timestep_embedding(timesteps, dim, flip_sin_to_cos=False, downscale_freq_shift=1.0, max_period=10000):
    import math
    half = dim // 2
    exponent = -math.log(max_period) * torch.arange(half, dtype=torch.float32) / (half - downscale_freq_shift)
    emb = timesteps[:, None].float() * torch.exp(exponent)[None, :]
    emb = torch.cat([torch.sin(emb), torch.cos(emb)], dim=-1)
    if flip_sin_to_cos:
        emb = torch.cat([emb[:, half:], emb[:, :half]], dim=-1)
    return emb


class _TimestepEmbedding(nn.Module):
    def __init__(self, in_channels, time_embed_dim):
        super().__init__()
        self.linear_1 = nn.Linear(in_channels, time_embed_dim)
        self.act = nn.SiLU()
        self.linear_2 = nn.Linear(time_embed_dim, time_embed_dim)

    def forward(self, sample):
        return self.linear_2(self.act(self.linear_1(sample)))


class _PixArtAlphaTextProjection(nn.Module):
    def __init__(self, in_features, hidden_size, out_features=None, act_fn="gelu_tanh"):
        super().__init__()
        self.linear_1 = nn.Linear(in_features, hidden_size)
        self.act_1 = {"gelu_tanh": nn.GELU(approximate="tanh"), "silu": nn.SiLU()}[act_fn]
        self.linear_2 = nn.Linear(hidden_size, out_features or hidden_size)

    def forward(self, caption):
        return self.linear_2(self.act_1(self.linear_1(caption)))


class _CombinedTimestepTextProjEmbeddings(nn.Module):
    def __init__(self, embedding_dim, pooled_projection_dim):
        super().__init__()
        self.timestep_embedder = _TimestepEmbedding(256, embedding_dim)
        self.text_embedder = _PixArtAlphaTextProjection(pooled_projection_dim, embedding_dim, act_fn="silu")

    def forward(self, timestep, pooled_projection):
        t = self.timestep_embedder(_get_timestep_embedding(timestep, 256, True, 0).to(pooled_projection.dtype))
        return t + self.text_embedder(pooled_projection)


class _CombinedTimestepGuidanceTextProjEmbeddings(nn.Module):
    def __init__(self, embedding_dim, pooled_projection_dim):
        super().__init__()
        self.timestep_embedder = _TimestepEmbedding(256, embedding_dim)
        self.guidance_embedder = _TimestepEmbedding(256, embedding_dim)
        self.text_embedder = _PixArtAlphaTextProjection(pooled_projection_dim, embedding_dim, act_fn="silu")

    def forward(self, timestep, guidance, pooled_projection):
        t = self.timestep_embedder(_get_timestep_embedding(timestep, 256, True, 0).to(pooled_projection.dtype))
        g = self.guidance_embedder(_get_timestep_embedding(guidance, 256, True, 0).to(pooled_projection.dtype))
        return t + g + self.text_embedder(pooled_projection)


def _get_1d_sincos_pos_embed_from_grid(embed_dim, pos):
    import numpy as np
    omega = np.arange(embed_dim // 2, dtype=np.float64)
    omega /= embed_dim / 2.0
    omega = 1.0 / 10000 ** omega
    out = np.einsum("m,d->md", pos.reshape(-1), omega)
    return np.concatenate([np.sin(out), np.cos(out)], axis=1)


def _get_2d_sincos_pos_embed(embed_dim, grid_size, base_size=16, interpolation_scale=1.0):
    import numpy as np
    if isinstance(grid_size, int):
        grid_size = (grid_size, grid_size)
    grid_h = np.arange(grid_size[0], dtype=np.float32) / (grid_size[0] / base_size) / interpolation_scale
    grid_w = np.arange(grid_size[1], dtype=np.float32) / (grid_size[1] / base_size) / interpolation_scale
    grid = np.meshgrid(grid_w, grid_h)  # here w goes first
    grid = np.stack(grid, axis=0)
    grid = grid.reshape([2, 1, grid_size[1], grid_size[0]])
    emb_h = _get_1d_sincos_pos_embed_from_grid(embed_dim // 2, grid[0])
    emb_w = _get_1d_sincos_pos_embed_from_grid(embed_dim // 2, grid[1])
    return np.concatenate([emb_h, emb_w], axis=1)


class _PatchEmbed(nn.Module):
    """embeddings.PatchEmbed (sincos positional table, no layer norm)."""

    def __init__(self, height=224, width=224, patch_size=16, in_channels=3, embed_dim=768, layer_norm=False, flatten=True,
                 bias=True, interpolation_scale=1, pos_embed_type="sincos", pos_embed_max_size=None):
        super().__init__()
        self.proj = nn.Conv2d(in_channels, embed_dim, kernel_size=(patch_size, patch_size), stride=patch_size, bias=bias)
        self.patch_size = patch_size
        self.height, self.width = height // patch_size, width // patch_size
        self.base_size = height // patch_size
        self.interpolation_scale = interpolation_scale
        num_patches = (height // patch_size) * (width // patch_size)
        pe = _get_2d_sincos_pos_embed(embed_dim, int(num_patches ** 0.5), base_size=self.base_size,
                                      interpolation_scale=interpolation_scale)
        self.register_buffer("pos_embed", torch.from_numpy(pe).float().unsqueeze(0), persistent=False)

    def forward(self, latent):
        height, width = latent.shape[-2] // self.patch_size, latent.shape[-1] // self.patch_size
        latent = self.proj(latent).flatten(2).transpose(1, 2)
        if self.height != height or self.width != width:
            pe = _get_2d_sincos_pos_embed(self.pos_embed.shape[-1], (height, width), base_size=self.base_size,
                                          interpolation_scale=self.interpolation_scale)
            pos_embed = torch.from_numpy(pe).float().unsqueeze(0)
        else:
            pos_embed = self.pos_embed
        return (latent + pos_embed).to(latent.dtype)


class _PixArtAlphaCombinedTimestepSizeEmbeddings(nn.Module):
    def __init__(self, embedding_dim, size_emb_dim, use_additional_conditions=False):
        super().__init__()
        assert not use_additional_conditions, "scaffolding covers use_additional_conditions=False (resolution=None call site)"
        self.timestep_embedder = _TimestepEmbedding(256, embedding_dim)

    def forward(self, timestep, resolution=None, aspect_ratio=None, batch_size=None, hidden_dtype=None):
        return self.timestep_embedder(_get_timestep_embedding(timestep, 256, True, 0).to(hidden_dtype))


class _AdaLayerNormSingle(nn.Module):
    def __init__(self, embedding_dim, use_additional_conditions=False):
        super().__init__()
        self.emb = _PixArtAlphaCombinedTimestepSizeEmbeddings(embedding_dim, embedding_dim // 3, use_additional_conditions)
        self.silu = nn.SiLU()
        self.linear = nn.Linear(embedding_dim, 6 * embedding_dim, bias=True)

    def forward(self, timestep, added_cond_kwargs=None, batch_size=None, hidden_dtype=None):
        embedded_timestep = self.emb(timestep, **(added_cond_kwargs or {}), batch_size=batch_size, hidden_dtype=hidden_dtype)
        return self.linear(self.silu(embedded_timestep)), embedded_timestep


class _FluxPosEmbed(nn.Module):
    def __init__(self, theta, axes_dim):
        super().__init__()
        self.theta = theta
        self.axes_dim = axes_dim

    def forward(self, ids):
        cos_out, sin_out = [], []
        pos = ids.float()
        for i in range(ids.shape[-1]):
            d = self.axes_dim[i]
            freqs = 1.0 / (self.theta ** (torch.arange(0, d, 2, dtype=torch.float64)[: d // 2] / d))
            ang = torch.outer(pos[:, i].to(torch.float64), freqs)
            cos_out.append(ang.cos().repeat_interleave(2, dim=1).float())
            sin_out.append(ang.sin().repeat_interleave(2, dim=1).float())
        return torch.cat(cos_out, dim=-1), torch.cat(sin_out, dim=-1)


def _apply_rotary_emb(x, freqs_cis, use_real=True, use_real_unbind_dim=-1):
    cos, sin = freqs_cis
    cos, sin = cos[None, None].to(x.device), sin[None, None].to(x.device)
    x_real, x_imag = x.reshape(*x.shape[:-1], -1, 2).unbind(-1)
    x_rotated = torch.stack([-x_imag, x_real], dim=-1).flatten(3)
    return (x.float() * cos + x_rotated.float() * sin).to(x.dtype)


_installed = False


def install():
    """Create the scaffolding packages and load the reference modules. Idempotent."""
    global _installed
    if _installed:
        return
    if not available():
        raise RuntimeError("reference tree not present: " + _FEATURE)

    def deprecate(*a, **k):
        pass

    logging = types.SimpleNamespace(get_logger=lambda n=None: _pylogging.getLogger(n or "diffusers"))

    def is_torch_version(op, ver):
        from packaging.version import parse
        import operator
        ops = {">=": operator.ge, ">": operator.gt, "<": operator.lt, "<=": operator.le, "==": operator.eq}
        return ops[op](parse(torch.__version__.split("+")[0]), parse(ver))

    ident = lambda cls: cls
    false = lambda *a, **k: False

    _pkg("diffusers")
    _pkg("diffusers.utils", deprecate=deprecate, logging=logging, is_torch_version=is_torch_version,
         is_torch_xla_available=false, USE_PEFT_BACKEND=False, BaseOutput=object,
         scale_lora_layers=deprecate, unscale_lora_layers=deprecate)
    _mod("diffusers.utils.torch_utils", maybe_allow_in_graph=ident, is_torch_version=is_torch_version)
    _mod("diffusers.utils.import_utils", is_torch_npu_available=false, is_torch_xla_version=false,
         is_xformers_available=false, is_torch_version=is_torch_version)
    _mod("diffusers.image_processor", IPAdapterMaskProcessor=type("IPAdapterMaskProcessor", (), {}))
    _mod("diffusers.configuration_utils", register_to_config=_register_to_config,
         ConfigMixin=type("ConfigMixin", (), {}), LegacyConfigMixin=type("LegacyConfigMixin", (), {}))
    _pkg("diffusers.models")
    _mod("diffusers.models.activations", get_activation=_get_activation, GEGLU=_GEGLU, GELU=_GELU,
         **{n: _placeholder(n) for n in ("ApproximateGELU", "FP32SiLU", "LinearActivation", "SwiGLU")})
    _mod("diffusers.models.embeddings", PixArtAlphaTextProjection=_PixArtAlphaTextProjection,
         CombinedTimestepGuidanceTextProjEmbeddings=_CombinedTimestepGuidanceTextProjEmbeddings,
         CombinedTimestepTextProjEmbeddings=_CombinedTimestepTextProjEmbeddings, FluxPosEmbed=_FluxPosEmbed,
         apply_rotary_emb=_apply_rotary_emb, PatchEmbed=_PatchEmbed,
         **{n: _placeholder(n) for n in ("SinusoidalPositionalEmbedding", "ImagePositionalEmbeddings")})
    _mod("diffusers.models.normalization", AdaLayerNormContinuous=_AdaLayerNormContinuous,
         AdaLayerNormZero=_AdaLayerNormZero, AdaLayerNormZeroSingle=_AdaLayerNormZeroSingle, RMSNorm=_RMSNorm,
         AdaLayerNormSingle=_AdaLayerNormSingle,
         **{n: _placeholder(n) for n in ("AdaGroupNorm", "AdaLayerNorm", "SD35AdaLayerNormZeroX", "FP32LayerNorm", "LpNorm")})
    _mix = lambda n: type(n, (), {})
    _mod("diffusers.loaders", FluxTransformer2DLoadersMixin=_mix("FluxTransformer2DLoadersMixin"),
         FromOriginalModelMixin=_mix("FromOriginalModelMixin"), PeftAdapterMixin=_mix("PeftAdapterMixin"))
    class _T2DOut:
        def __init__(self, sample=None):
            self.sample = sample
    _mod("diffusers.models.modeling_outputs", Transformer2DModelOutput=_T2DOut)
    _mod("diffusers.models.modeling_utils", LegacyModelMixin=nn.Module, ModelMixin=nn.Module)
    _pkg("diffusers.models.transformers")

    # torchvision.transforms.functional.normalize (components/feature_extractor.py:6,56)
    if "torchvision" not in sys.modules:
        def tv_normalize(t, mean, std, inplace=False):
            return (t.clone() - mean) / std
        _pkg("torchvision"); _pkg("torchvision.transforms")
        _mod("torchvision.transforms.functional", normalize=tv_normalize)
        sys.modules["torchvision.transforms"].functional = sys.modules["torchvision.transforms.functional"]

    def load(modname, relpath):
        spec = importlib.util.spec_from_file_location(modname, os.path.join(_FEATURE, relpath))
        m = importlib.util.module_from_spec(spec)
        sys.modules[modname] = m
        spec.loader.exec_module(m)
        return m

    # dependency order
    load("diffusers.models.attention_processor", "diffusers/models/attention_processor.py")
    load("diffusers.models.upsampling", "diffusers/models/upsampling.py")
    load("diffusers.models.downsampling", "diffusers/models/downsampling.py")
    load("diffusers.models.resnet", "diffusers/models/resnet.py")
    load("diffusers.models.attention", "diffusers/models/attention.py")
    load("diffusers.models.transformers.transformer_2d", "diffusers/models/transformers/transformer_2d.py")
    load("diffusers.models.transformers.transformer_flux", "diffusers/models/transformers/transformer_flux.py")
    load("gdf_ref_feature_extractor", "components/feature_extractor.py")
    # components/attention.py imports names from the installed diffusers package root
    sys.modules["diffusers.models.attention_processor"].__dict__.setdefault("AttnProcessor2_0", None)
    _installed = True


def modules():
    install()
    m = sys.modules
    ns = types.SimpleNamespace(
        ResnetBlock2D=m["diffusers.models.resnet"].ResnetBlock2D,
        Upsample2D=m["diffusers.models.upsampling"].Upsample2D,
        Downsample2D=m["diffusers.models.downsampling"].Downsample2D,
        BasicTransformerBlock=m["diffusers.models.attention"].BasicTransformerBlock,
        FeedForward=m["diffusers.models.attention"].FeedForward,
        Attention=m["diffusers.models.attention_processor"].Attention,
        AttnProcessor=m["diffusers.models.attention_processor"].AttnProcessor,
        AttnProcessor2_0=m["diffusers.models.attention_processor"].AttnProcessor2_0,
        Transformer2DModel=m["diffusers.models.transformers.transformer_2d"].Transformer2DModel,
        FluxTransformer2DModel=m["diffusers.models.transformers.transformer_flux"].FluxTransformer2DModel,
        FeatureStore=m["gdf_ref_feature_extractor"].FeatureStore,
        FeatureGatherer=m["gdf_ref_feature_extractor"].FeatureGatherer,
    )
    return ns


def flux_attn_store_processor():
    """components/attention.py::FluxAttnStoreProcessor (the eager MMDiT processor behind `self-map` / `cross-map`)."""
    attn_store_processor()
    return sys.modules["gdf_ref_attention"].FluxAttnStoreProcessor


def attn_store_processor():
    """components/attention.py::AttnStoreProcessor (the eager '-map' processor)."""
    install()
    if "gdf_ref_attention" not in sys.modules:
        # its module-level imports: `from diffusers.models.attention_processor import ...`, einops, PIL/cv2 are lazy
        path = os.path.join(_FEATURE, "components", "attention.py")
        src = open(path).read()
        spec = importlib.util.spec_from_file_location("gdf_ref_attention", path)
        mod = importlib.util.module_from_spec(spec)
        sys.modules["gdf_ref_attention"] = mod
        try:
            spec.loader.exec_module(mod)
        except Exception as e:  # missing optional third-party imports at module top
            raise RuntimeError(f"cannot import reference components/attention.py: {e}; head:\n{src[:600]}")
    return sys.modules["gdf_ref_attention"].AttnStoreProcessor
