"""CPU ORACLE helper (test infrastructure, build container only): the HYBRID whole-UNet oracle of SURVEY.md §8(c)(4).

Imports the reference's OWN `UNet2DConditionModel` (/root/reference/feature/diffusers/models/unet/unet_2d_condition.py —
`__init__` :171-484, `forward` :1040-1319: time / text_time embedding path, skip stack, mid block, `conv_norm_out`, the
`unet-in / after-conv-in / out` gather sites) and the reference's OWN `prepare_feature_extractor`
(feature/components/feature_extractor.py:92-288: the hook-id scheme) and lets them drive the reference's own
ResnetBlock2D / Transformer2DModel / Down- / Upsample2D.  What diffusers==0.32.2 does NOT vendor into the reference tree is
restated here from the published algorithm and marked [restated]:
  * `unets/unet_2d_blocks.py`: CrossAttnDownBlock2D / DownBlock2D / UNetMidBlock2DCrossAttn / CrossAttnUpBlock2D / UpBlock2D
    containers (ModuleLists `resnets`, `attentions`, `downsamplers`, `upsamplers`; their forward loops) and the
    get_down_block / get_mid_block / get_up_block factories;
  * `embeddings.py`: Timesteps, TimestepEmbedding (the other embedding classes the file imports are placeholders: unused by
    the SD1.5 / SD2.1 / SDXL configurations).
Nothing here is copied from the reference and nothing here travels to the GPU box: tests/golden/gen_golden_unet.py runs it
once and commits the resulting input / output vectors (tests/golden/unet_tiny_*.npz).
"""
import importlib.util
import os
import sys

import torch
import torch.nn as nn

from . import ref_blocks as RB


class _Timesteps(nn.Module):
    """[restated] diffusers==0.32.2 embeddings.Timesteps"""

    def __init__(self, num_channels, flip_sin_to_cos, downscale_freq_shift, scale=1):
        super().__init__()
        self.num_channels, self.flip, self.shift, self.scale = num_channels, flip_sin_to_cos, downscale_freq_shift, scale

    def forward(self, timesteps):
        return RB._get_timestep_embedding(timesteps, self.num_channels, flip_sin_to_cos=self.flip, downscale_freq_shift=self.shift)


class _TimestepEmbedding(nn.Module):
    """[restated] diffusers==0.32.2 embeddings.TimestepEmbedding (no cond_proj / post_act: unused by SD / SDXL configs)"""

    def __init__(self, in_channels, time_embed_dim, act_fn="silu", out_dim=None, post_act_fn=None, cond_proj_dim=None,
                 sample_proj_bias=True):
        super().__init__()
        assert act_fn == "silu" and post_act_fn is None and cond_proj_dim is None
        self.linear_1 = nn.Linear(in_channels, time_embed_dim, sample_proj_bias)
        self.act = nn.SiLU()
        self.linear_2 = nn.Linear(time_embed_dim, out_dim or time_embed_dim, sample_proj_bias)

    def forward(self, sample, condition=None):
        return self.linear_2(self.act(self.linear_1(sample)))


def _blocks_module(M):
    """[restated] diffusers==0.32.2 models/unets/unet_2d_blocks.py, the five block containers of SD1.5 / SD2.1 / SDXL"""
    Res, T2D, Down, Up = M.ResnetBlock2D, M.Transformer2DModel, M.Downsample2D, M.Upsample2D

    def res(ci, co, temb, eps, groups, dropout=0.0):
        return Res(in_channels=ci, out_channels=co, temb_channels=temb, eps=eps, groups=groups, dropout=dropout,
                   time_embedding_norm="default", non_linearity="silu", output_scale_factor=1.0, pre_norm=True)

    def vit(c, heads, layers, cross, groups, linear):
        return T2D(heads, c // heads, in_channels=c, num_layers=layers, cross_attention_dim=cross, norm_num_groups=groups,
                   use_linear_projection=linear, only_cross_attention=False, upcast_attention=False, attention_type="default")

    class DownBlock2D(nn.Module):
        def __init__(self, num_layers, in_channels, out_channels, temb_channels, add_downsample, resnet_eps, resnet_groups,
                     downsample_padding, **_):
            super().__init__()
            self.resnets = nn.ModuleList([res(in_channels if i == 0 else out_channels, out_channels, temb_channels, resnet_eps,
                                              resnet_groups) for i in range(num_layers)])
            self.downsamplers = nn.ModuleList([Down(out_channels, use_conv=True, out_channels=out_channels,
                                                    padding=downsample_padding, name="op")]) if add_downsample else None

        def forward(self, hidden_states, temb=None, **_):
            outs = ()
            for r in self.resnets:
                hidden_states = r(hidden_states, temb)
                outs += (hidden_states,)
            if self.downsamplers is not None:
                for d in self.downsamplers:
                    hidden_states = d(hidden_states)
                outs += (hidden_states,)
            return hidden_states, outs

    class CrossAttnDownBlock2D(DownBlock2D):
        has_cross_attention = True

        def __init__(self, num_layers, transformer_layers_per_block, out_channels, cross_attention_dim, num_attention_heads,
                     resnet_groups, use_linear_projection, **kw):
            super().__init__(num_layers=num_layers, out_channels=out_channels, resnet_groups=resnet_groups, **kw)
            tl = transformer_layers_per_block
            tl = [tl] * num_layers if isinstance(tl, int) else tl
            self.attentions = nn.ModuleList([vit(out_channels, num_attention_heads, tl[i], cross_attention_dim, resnet_groups,
                                                 use_linear_projection) for i in range(num_layers)])

        def forward(self, hidden_states, temb=None, encoder_hidden_states=None, attention_mask=None, cross_attention_kwargs=None,
                    encoder_attention_mask=None, additional_residuals=None):
            outs = ()
            for r, a in zip(self.resnets, self.attentions):
                hidden_states = r(hidden_states, temb)
                hidden_states = a(hidden_states, encoder_hidden_states=encoder_hidden_states, return_dict=False)[0]
                outs += (hidden_states,)
            if self.downsamplers is not None:
                for d in self.downsamplers:
                    hidden_states = d(hidden_states)
                outs += (hidden_states,)
            return hidden_states, outs

    class UNetMidBlock2DCrossAttn(nn.Module):
        has_cross_attention = True

        def __init__(self, in_channels, temb_channels, resnet_eps, resnet_groups, transformer_layers_per_block,
                     num_attention_heads, cross_attention_dim, use_linear_projection, **_):
            super().__init__()
            tl = transformer_layers_per_block
            tl = tl[0] if isinstance(tl, (list, tuple)) else tl
            self.resnets = nn.ModuleList([res(in_channels, in_channels, temb_channels, resnet_eps, resnet_groups) for _ in range(2)])
            self.attentions = nn.ModuleList([vit(in_channels, num_attention_heads, tl, cross_attention_dim, resnet_groups,
                                                 use_linear_projection)])

        def forward(self, hidden_states, temb=None, encoder_hidden_states=None, **_):
            hidden_states = self.resnets[0](hidden_states, temb)
            for a, r in zip(self.attentions, self.resnets[1:]):
                hidden_states = a(hidden_states, encoder_hidden_states=encoder_hidden_states, return_dict=False)[0]
                hidden_states = r(hidden_states, temb)
            return hidden_states

    class UpBlock2D(nn.Module):
        def __init__(self, num_layers, in_channels, out_channels, prev_output_channel, temb_channels, add_upsample, resnet_eps,
                     resnet_groups, **_):
            super().__init__()
            rs = []
            for i in range(num_layers):
                skip = in_channels if i == num_layers - 1 else out_channels
                cin = prev_output_channel if i == 0 else out_channels
                rs.append(res(cin + skip, out_channels, temb_channels, resnet_eps, resnet_groups))
            self.resnets = nn.ModuleList(rs)
            self.upsamplers = nn.ModuleList([Up(out_channels, use_conv=True, out_channels=out_channels)]) if add_upsample else None

        def _cat(self, hidden_states, res_tuple):
            return torch.cat([hidden_states, res_tuple[-1]], dim=1), res_tuple[:-1]

        def forward(self, hidden_states, res_hidden_states_tuple, temb=None, upsample_size=None, **_):
            for r in self.resnets:
                hidden_states, res_hidden_states_tuple = self._cat(hidden_states, res_hidden_states_tuple)
                hidden_states = r(hidden_states, temb)
            if self.upsamplers is not None:
                for u in self.upsamplers:
                    hidden_states = u(hidden_states, upsample_size)
            return hidden_states

    class CrossAttnUpBlock2D(UpBlock2D):
        has_cross_attention = True

        def __init__(self, num_layers, transformer_layers_per_block, out_channels, cross_attention_dim, num_attention_heads,
                     resnet_groups, use_linear_projection, **kw):
            super().__init__(num_layers=num_layers, out_channels=out_channels, resnet_groups=resnet_groups, **kw)
            tl = transformer_layers_per_block
            tl = [tl] * num_layers if isinstance(tl, int) else tl
            self.attentions = nn.ModuleList([vit(out_channels, num_attention_heads, tl[i], cross_attention_dim, resnet_groups,
                                                 use_linear_projection) for i in range(num_layers)])

        def forward(self, hidden_states, res_hidden_states_tuple, temb=None, encoder_hidden_states=None, upsample_size=None, **_):
            for r, a in zip(self.resnets, self.attentions):
                hidden_states, res_hidden_states_tuple = self._cat(hidden_states, res_hidden_states_tuple)
                hidden_states = r(hidden_states, temb)
                hidden_states = a(hidden_states, encoder_hidden_states=encoder_hidden_states, return_dict=False)[0]
            if self.upsamplers is not None:
                for u in self.upsamplers:
                    hidden_states = u(hidden_states, upsample_size)
            return hidden_states

    kinds = dict(DownBlock2D=DownBlock2D, CrossAttnDownBlock2D=CrossAttnDownBlock2D, UpBlock2D=UpBlock2D,
                 CrossAttnUpBlock2D=CrossAttnUpBlock2D, UNetMidBlock2DCrossAttn=UNetMidBlock2DCrossAttn)

    def get_down_block(down_block_type, **kw):
        return kinds[down_block_type](**kw)

    def get_mid_block(mid_block_type, **kw):
        return kinds[mid_block_type](**kw)

    def get_up_block(up_block_type, **kw):
        return kinds[up_block_type](**kw)

    return dict(get_down_block=get_down_block, get_mid_block=get_mid_block, get_up_block=get_up_block)


_cls = None


def reference_unet_class():
    """-> (UNet2DConditionModel of the reference tree, reference prepare_feature_extractor, FeatureStore)"""
    global _cls
    if _cls is not None:
        return _cls
    RB.install()
    M = RB.modules()
    ph = RB._placeholder
    emb = sys.modules["diffusers.models.embeddings"]
    emb.Timesteps, emb.TimestepEmbedding = _Timesteps, _TimestepEmbedding
    for n in ("GaussianFourierProjection", "GLIGENTextBoundingboxProjection", "ImageHintTimeEmbedding", "ImageProjection",
              "ImageTimeEmbedding", "TextImageProjection", "TextImageTimeEmbedding", "TextTimeEmbedding"):
        if not hasattr(emb, n):
            setattr(emb, n, ph(n))
    ld = sys.modules["diffusers.loaders"]
    ld.UNet2DConditionLoadersMixin = type("UNet2DConditionLoadersMixin", (), {})
    ld.__path__ = []
    RB._mod("diffusers.loaders.single_file_model", FromOriginalModelMixin=ld.FromOriginalModelMixin)
    RB._pkg("diffusers.models.unet")
    RB._mod("diffusers.models.unet.unet_2d_blocks", **_blocks_module(M))
    path = os.path.join(RB._FEATURE, "diffusers/models/unet/unet_2d_condition.py")
    spec = importlib.util.spec_from_file_location("diffusers.models.unet.unet_2d_condition", path)
    mod = importlib.util.module_from_spec(spec)
    sys.modules["diffusers.models.unet.unet_2d_condition"] = mod
    spec.loader.exec_module(mod)
    fe = sys.modules["gdf_ref_feature_extractor"]
    _cls = (mod.UNet2DConditionModel, fe.prepare_feature_extractor, fe.FeatureStore)
    return _cls


def build_reference_unet(arch):
    """Instantiate the reference UNet2DConditionModel for an oracle architecture dict (oracle/unet_ref.py ARCHS / tiny_arch)."""
    UNet, _, _ = reference_unet_class()
    L = len(arch["block_out_channels"])
    down = tuple("CrossAttnDownBlock2D" if a else "DownBlock2D" for a in arch["down_attn"])
    up = tuple("CrossAttnUpBlock2D" if a else "UpBlock2D" for a in reversed(arch["down_attn"]))
    kw = dict(in_channels=arch["in_channels"], out_channels=arch["out_channels"], down_block_types=down, up_block_types=up,
              block_out_channels=tuple(arch["block_out_channels"]), layers_per_block=arch["layers_per_block"],
              cross_attention_dim=arch["cross_dim"], transformer_layers_per_block=list(arch["transformer_layers"]),
              attention_head_dim=tuple(arch["heads"]), use_linear_projection=bool(arch["linear_proj"]))
    if arch["addition_embed"] == "text_time":
        kw.update(addition_embed_type="text_time", addition_time_embed_dim=arch["addition_time_embed_dim"],
                  projection_class_embeddings_input_dim=arch["add_in_dim"])
    assert L == len(down)
    return UNet(**kw)
