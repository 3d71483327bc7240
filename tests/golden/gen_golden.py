"""Generate golden vectors from the REFERENCE's own block modules (build container only).

    python tests/golden/gen_golden.py        # needs /root/reference; writes tests/golden/*.npz

Each fixture is pure data: fp16-representable weights + inputs (so the file is self-contained)
and the fp32 outputs / hook tensors produced by the reference modules
(feature/diffusers/models/{resnet,attention,attention_processor,upsampling,downsampling}.py,
feature/diffusers/models/transformers/transformer_2d.py, feature/components/feature_extractor.py,
feature/components/attention.py) imported from /root/reference by oracle/ref_blocks.py.
"""
import math
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle import ref_blocks as RB  # noqa: E402


def init_module(mod, gen):
    """Same init rule as oracle.unet_ref.synth_params, applied to a reference nn.Module."""
    sd = {}
    for name, p in mod.state_dict().items():
        is_norm = "norm" in name
        if name.endswith("weight") and not is_norm:
            fan_in = int(np.prod(p.shape[1:]))
            w = torch.randn(p.shape, generator=gen) / math.sqrt(fan_in)
        elif name.endswith("weight"):
            w = 1.0 + 0.1 * torch.randn(p.shape, generator=gen)
        elif is_norm:
            w = 0.1 * torch.randn(p.shape, generator=gen)
        else:
            w = 0.05 * torch.randn(p.shape, generator=gen)
        sd[name] = w.half().float()
    mod.load_state_dict(sd)
    return sd


def attach(m, mod, module_id, store):
    mod.feature_gatherer = m.FeatureGatherer(module_id, store)


def save(name, weights, inputs, outputs, meta):
    arrs = {}
    for k, v in weights.items():
        arrs["w:" + k] = v.half().numpy()
    for k, v in inputs.items():
        arrs["in:" + k] = v.numpy().astype(np.float32)
    for k, v in outputs.items():
        arrs["out:" + k] = v.detach().float().numpy()
    arrs["meta"] = np.array(repr(meta))
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **arrs)
    print(name, "->", os.path.getsize(path) // 1024, "KiB", list(outputs.keys()))


def randh(gen, *shape, scale=1.0):
    return (scale * torch.randn(*shape, generator=gen)).half().float()


@torch.no_grad()
def main():
    m = RB.modules()
    torch.manual_seed(0)

    # ---- ResnetBlock2D (resnet.py:189-379), with and without 1x1 shortcut -------------------
    for name, cin, cout, hw in (("resnet_same", 32, 32, 8), ("resnet_shortcut", 96, 64, 6)):
        g = torch.Generator().manual_seed(11)
        store = m.FeatureStore({}, 1, True)            # accept-all, train_unet=True -> no .to('cuda')
        mod = m.ResnetBlock2D(in_channels=cin, out_channels=cout, temb_channels=48, groups=32, eps=1e-5)
        w = init_module(mod, g)
        attach(m, mod, "blk-res", store)
        x, temb = randh(g, 2, cin, hw, hw, scale=1.5), randh(g, 2, 48)
        y = mod(x, temb)
        outs = {"y": y}
        outs.update({"hook:" + k: v for k, v in store.stored_feats.items()})
        save(name, w, {"x": x, "temb": temb}, outs, dict(cin=cin, cout=cout, hw=hw, eps=1e-5))

    # ---- Downsample2D / Upsample2D ------------------------------------------------------------
    g = torch.Generator().manual_seed(12)
    store = m.FeatureStore({}, 1, True)
    mod = m.Downsample2D(32, use_conv=True, out_channels=32, padding=1, name="op")
    w = init_module(mod, g); attach(m, mod, "blk-downsampler", store)
    x = randh(g, 2, 32, 8, 8)
    y = mod(x)
    save("downsample", w, {"x": x}, {"y": y, **{"hook:" + k: v for k, v in store.stored_feats.items()}}, {})
    store = m.FeatureStore({}, 1, True)
    mod = m.Upsample2D(32, use_conv=True, out_channels=32)
    w = init_module(mod, g); attach(m, mod, "blk-upsampler", store)
    x = randh(g, 2, 32, 5, 5)
    y = mod(x)
    save("upsample", w, {"x": x}, {"y": y, **{"hook:" + k: v for k, v in store.stored_feats.items()}}, {})

    # ---- Transformer2DModel (+BasicTransformerBlock, Attention, FeedForward) --------------------
    def vit_case(name, c, heads, hw, depth, cross, linear, use_map, resize_ratio=1, seed=13):
        g = torch.Generator().manual_seed(seed)
        store = m.FeatureStore({}, resize_ratio, True)
        mod = m.Transformer2DModel(num_attention_heads=heads, attention_head_dim=c // heads, in_channels=c,
                                   num_layers=depth, cross_attention_dim=cross, norm_num_groups=32,
                                   use_linear_projection=linear)
        w = init_module(mod, g)
        # same gatherer assignment as components/feature_extractor.py:131-157
        attach(m, mod, "blk-vit", store)
        for i, bb in enumerate(mod.transformer_blocks):
            attach(m, bb, f"blk-vit-block{i}", store)
            attach(m, bb.attn1, f"blk-vit-block{i}-self", store)
            attach(m, bb.attn2, f"blk-vit-block{i}-cross", store)
            attach(m, bb.ff, f"blk-vit-block{i}-ffn", store)
            if use_map:
                P = RB.attn_store_processor()
                bb.attn1.set_processor(P(None, "down")); bb.attn2.set_processor(P(None, "down"))
        x, ctx = randh(g, 2, c, hw, hw, scale=2.0), randh(g, 2, 77, cross)
        y = mod(x, encoder_hidden_states=ctx, return_dict=False)[0]
        outs = {"y": y}
        outs.update({"hook:" + k: v for k, v in store.stored_feats.items()})
        save(name, w, {"x": x, "ctx": ctx}, outs,
             dict(c=c, heads=heads, hw=hw, depth=depth, cross=cross, linear=linear, use_map=use_map,
                  resize_ratio=resize_ratio, order=list(store.stored_feats.keys())))

    vit_case("vit_linear_sdpa", 64, 2, 6, 2, 48, True, False)          # SDXL style: linear proj, d=32
    vit_case("vit_linear_d64", 128, 2, 4, 1, 32, True, False)          # SDXL head dim 64
    vit_case("vit_conv_map", 160, 4, 4, 1, 40, False, True)            # SD1.5 style: conv proj, d=40, '-map' hooks
    vit_case("vit_linear_resize2", 64, 2, 8, 1, 48, True, False, resize_ratio=2)   # feature_resize pooling


if __name__ == "__main__":
    if not RB.available():
        sys.exit("reference tree not found; goldens can only be generated in the build container")
    main()
