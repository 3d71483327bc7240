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
    _mod("diffusers.models.activations", get_activation=_get_activation, GEGLU=_GEGLU,
         **{n: _placeholder(n) for n in ("GELU", "ApproximateGELU", "FP32SiLU", "LinearActivation", "SwiGLU")})
    _mod("diffusers.models.embeddings",
         **{n: _placeholder(n) for n in ("SinusoidalPositionalEmbedding", "ImagePositionalEmbeddings", "PatchEmbed",
                                         "PixArtAlphaTextProjection")})
    _mod("diffusers.models.normalization",
         **{n: _placeholder(n) for n in ("AdaGroupNorm", "AdaLayerNorm", "AdaLayerNormContinuous", "AdaLayerNormZero",
                                         "RMSNorm", "SD35AdaLayerNormZeroX", "AdaLayerNormSingle", "FP32LayerNorm", "LpNorm")})
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
        FeatureStore=m["gdf_ref_feature_extractor"].FeatureStore,
        FeatureGatherer=m["gdf_ref_feature_extractor"].FeatureGatherer,
    )
    return ns


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
