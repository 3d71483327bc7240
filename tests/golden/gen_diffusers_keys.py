"""Parameter-name / shape / config fixtures for tests/fake_diffusers (build container only).

    python tests/golden/gen_diffusers_keys.py      # needs /root/reference; writes tests/golden/diffusers_keys.json

The real-checkpoint branch of the boundary (components/models.py get_diffusion_model without GDF_SYNTHETIC_WEIGHTS) loads
`pipe.unet.state_dict()` / `pipe.transformer.state_dict()` and reads `pipe.unet.config`.  Neither diffusers nor a checkpoint exists on
the build or GPU boxes, so tests/fake_diffusers hands out `nn.Module` shells that carry diffusers' PARAMETER NAMES, SHAPES and
CONFIG OBJECTS at the TRUE architectures.  Those come from here, i.e. from the reference's OWN model classes instantiated on the
`meta` device (no memory) with the constructor arguments of the published `config.json` files:

  * UNet2DConditionModel (/root/reference/feature/diffusers/models/unet/unet_2d_condition.py:171-484) for
    stable-diffusion-v1-5, stable-diffusion-2-1-base, stable-diffusion-xl-base-1.0  [config.json values: restated from the published files]
  * FluxTransformer2DModel (/root/reference/feature/diffusers/models/transformers/transformer_flux.py:233-411) for FLUX.1-dev
  * Transformer2DModel in its PixArt configuration (transformers/transformer_2d.py, `ada_norm_single` + patched input) for
    PixArt-Sigma-XL-2-1024-MS (diffusers==0.32.2 names the same module tree PixArtTransformer2DModel; that class is not vendored)

The fixture is pure data: {model: {"config": {...}, "keys": [[name, [shape]], ...]}}.  AutoencoderKL is not vendored in the reference
tree at all: its key list is restated inside tests/fake_diffusers from the published module tree.
"""
import json
import os
import sys

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle import ref_blocks as RB  # noqa: E402
from oracle import ref_unet as RU  # noqa: E402

# ---- unet/config.json of the three checkpoints (fields a file does not carry fall back to the class defaults, as in diffusers) ----
SD15 = dict(sample_size=64, in_channels=4, out_channels=4, center_input_sample=False, flip_sin_to_cos=True, freq_shift=0,
            down_block_types=("CrossAttnDownBlock2D", "CrossAttnDownBlock2D", "CrossAttnDownBlock2D", "DownBlock2D"),
            up_block_types=("UpBlock2D", "CrossAttnUpBlock2D", "CrossAttnUpBlock2D", "CrossAttnUpBlock2D"),
            block_out_channels=(320, 640, 1280, 1280), layers_per_block=2, downsample_padding=1, mid_block_scale_factor=1, act_fn="silu",
            norm_num_groups=32, norm_eps=1e-5, cross_attention_dim=768, attention_head_dim=8)
SD21 = dict(SD15, cross_attention_dim=1024, attention_head_dim=(5, 10, 20, 20), use_linear_projection=True, dual_cross_attention=False)
SDXL = dict(sample_size=128, in_channels=4, out_channels=4, center_input_sample=False, flip_sin_to_cos=True, freq_shift=0,
            down_block_types=("DownBlock2D", "CrossAttnDownBlock2D", "CrossAttnDownBlock2D"),
            up_block_types=("CrossAttnUpBlock2D", "CrossAttnUpBlock2D", "UpBlock2D"), block_out_channels=(320, 640, 1280),
            layers_per_block=2, downsample_padding=1, mid_block_scale_factor=1, act_fn="silu", norm_num_groups=32, norm_eps=1e-5,
            cross_attention_dim=2048, transformer_layers_per_block=[1, 2, 10], attention_head_dim=(5, 10, 20), use_linear_projection=True,
            addition_embed_type="text_time", addition_time_embed_dim=256, projection_class_embeddings_input_dim=2816,
            upcast_attention=None)
FLUX = dict(patch_size=1, in_channels=64, num_layers=19, num_single_layers=38, attention_head_dim=128, num_attention_heads=24,
            joint_attention_dim=4096, pooled_projection_dim=768, guidance_embeds=True, axes_dims_rope=(16, 56, 56))
PIXART_SIGMA = dict(num_attention_heads=16, attention_head_dim=72, in_channels=4, out_channels=8, num_layers=28, dropout=0.0,
                    norm_num_groups=32, cross_attention_dim=1152, attention_bias=True, sample_size=128, patch_size=2,
                    activation_fn="gelu-approximate", num_embeds_ada_norm=1000, upcast_attention=False, norm_type="ada_norm_single",
                    norm_elementwise_affine=False, norm_eps=1e-6, interpolation_scale=2, use_additional_conditions=False,
                    caption_channels=4096, attention_type="default")


def _jsonable(v):
    if isinstance(v, (tuple, list)):
        return [_jsonable(x) for x in v]
    if isinstance(v, dict):
        return {k: _jsonable(x) for k, x in v.items()}
    if isinstance(v, (int, float, str, bool)) or v is None:
        return v
    return repr(v)


def describe(net):
    return {"config": _jsonable(dict(net.config)), "keys": [[k, list(v.shape)] for k, v in net.state_dict().items()]}


def main():
    UNet, _, _ = RU.reference_unet_class()
    m = RB.modules()
    out = {}
    with torch.device("meta"):
        out["unet-1-5"] = describe(UNet(**SD15))
        out["unet-2-1"] = describe(UNet(**SD21))
        out["unet-xl"] = describe(UNet(**SDXL))
        out["flux"] = describe(m.FluxTransformer2DModel(**FLUX))
        pix = describe(m.Transformer2DModel(**PIXART_SIGMA))
        # `pos_embed.pos_embed` is a non-persistent buffer in diffusers' PatchEmbed (not in checkpoints or state_dict());
        # the oracle scaffolding registers it persistently
        pix["keys"] = [kv for kv in pix["keys"] if kv[0] != "pos_embed.pos_embed"]
        out["pixart-sigma"] = pix
    path = os.path.join(HERE, "diffusers_keys.json")
    with open(path, "w") as f:
        json.dump(out, f, separators=(",", ":"))
    for k, v in out.items():
        n = sum(int(torch.tensor(s).prod()) if s else 1 for _, s in v["keys"])
        print(f"{k:14s} {len(v['keys']):5d} tensors {n / 1e9:7.3f} G parameters")
    print(os.path.getsize(path) // 1024, "KiB")


if __name__ == "__main__":
    main()
