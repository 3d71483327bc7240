"""CPU ORACLE (test infrastructure only) — fp32 restatement of the PixArt (alpha / sigma) DiT single forward with the
reference's per-layer hooks (named in BASELINE.json's north_star as "PixArt-DiT").

This file is a CHECKER (tests/, smoke(), bench cpu_baseline only); the product path never imports it.

Restated from (paths relative to /root/reference/feature/):
  Transformer2DModel.forward, patched branch          diffusers/models/transformers/transformer_2d.py:404-475,
      _operate_on_patched_inputs :496-516, _get_output_for_patched_inputs :540-575
  BasicTransformerBlock.forward, norm_type == 'ada_norm_single'   diffusers/models/attention.py:469-592
      (scale_shift_table :498-503, attn1 gate :524, norm2 NOT applied before attn2 :541-543, ff modulate :570-583)
  Attention + AttnProcessor2_0 (bias, additive encoder mask)      diffusers/models/attention_processor.py:3244-3331
  FeedForward 'gelu-approximate'                                  diffusers/models/attention.py:1249-1258
  hook ids                                                        components/feature_extractor.py:250-286
  call site                                                       diffusion_feature.py:466-474
Un-vendored diffusers==0.32.2 pieces restated from the published algorithm: PatchEmbed + get_2d_sincos_pos_embed,
AdaLayerNormSingle / PixArtAlphaCombinedTimestepSizeEmbeddings (use_additional_conditions=False), Timesteps,
TimestepEmbedding, PixArtAlphaTextProjection(gelu_tanh), activations.GELU(tanh).

Parity pinning: tests/golden/pixart_tiny.npz = outputs of the reference's OWN Transformer2DModel / BasicTransformerBlock /
Attention / FeedForward / FeatureStore on a tiny DiT (tests/golden/gen_golden_pixart.py); the un-vendored classes above
come from the scaffolding in oracle/ref_blocks.py => for them "parity unpinned".
"""
import math
from collections import OrderedDict

import numpy as np
import torch
import torch.nn.functional as F

from .unet_ref import Store, timestep_sinusoid  # noqa: F401

# PixArt-Sigma-XL-2-1024-MS transformer/config.json (components/models.py:72-111 of the reference)
ARCH_PIXART_SIGMA = dict(num_attention_heads=16, attention_head_dim=72, in_channels=4, out_channels=8, num_layers=28,
                         patch_size=2, sample_size=128, caption_channels=4096, interpolation_scale=2)


def tiny_arch(heads=8, num_layers=2, caption_channels=128, sample_size=16):
    """heads x 72 must be a multiple of 64 for the native GEMMs (K tiles): 8 heads -> inner dim 576."""
    a = dict(ARCH_PIXART_SIGMA)
    a.update(num_attention_heads=heads, num_layers=num_layers, caption_channels=caption_channels, sample_size=sample_size,
             interpolation_scale=max(sample_size // 64, 1))
    return a


def inner_dim(arch):
    return arch["num_attention_heads"] * arch["attention_head_dim"]


def param_shapes(arch):
    C = inner_dim(arch); p = arch["patch_size"]
    S = OrderedDict()

    def lin(n, o, i):
        S[n + ".weight"] = (o, i); S[n + ".bias"] = (o,)

    S["pos_embed.proj.weight"] = (C, arch["in_channels"], p, p); S["pos_embed.proj.bias"] = (C,)
    lin("adaln_single.emb.timestep_embedder.linear_1", C, 256)
    lin("adaln_single.emb.timestep_embedder.linear_2", C, C)
    lin("adaln_single.linear", 6 * C, C)
    lin("caption_projection.linear_1", C, arch["caption_channels"])
    lin("caption_projection.linear_2", C, C)
    for i in range(arch["num_layers"]):
        b = f"transformer_blocks.{i}"
        S[b + ".scale_shift_table"] = (6, C)
        for a in ("attn1", "attn2"):
            for n in ("to_q", "to_k", "to_v", "to_out.0"):
                lin(f"{b}.{a}.{n}", C, C)
        lin(b + ".ff.net.0.proj", 4 * C, C); lin(b + ".ff.net.2", C, 4 * C)
    S["scale_shift_table"] = (2, C)
    lin("proj_out", p * p * arch["out_channels"], C)
    return S


def synth_params(arch, seed=0, dtype=torch.float32):
    g = torch.Generator().manual_seed(seed)
    P = OrderedDict()
    C = inner_dim(arch)
    for name, shape in param_shapes(arch).items():
        if name.endswith("scale_shift_table"):
            w = torch.randn(shape, generator=g) / C ** 0.5          # transformer_2d.py:304, attention.py:411
        elif name.endswith(".weight"):
            fan = 1
            for s in shape[1:]:
                fan *= s
            w = torch.randn(shape, generator=g) / math.sqrt(fan)
        else:
            w = 0.05 * torch.randn(shape, generator=g)
        P[name] = w.half().to(dtype)
    return P


def synth_inputs(arch, batch, lat, n_txt, seed=1, valid=None):
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(batch, arch["in_channels"], lat, lat, generator=g).half().float()
    ctx = torch.randn(batch, n_txt, arch["caption_channels"], generator=g).half().float()
    mask = torch.zeros(batch, n_txt)
    for b in range(batch):
        mask[b, : (valid[b] if valid else n_txt)] = 1
    return dict(hidden_states=x, encoder_hidden_states=ctx, timestep=torch.full((batch,), 100.0), encoder_attention_mask=mask)


def _lin(P, n, x):
    return F.linear(x, P[n + ".weight"], P.get(n + ".bias"))


def sincos_pos_embed(embed_dim, grid_h, grid_w, base_size, interpolation_scale):
    """embeddings.get_2d_sincos_pos_embed (+_from_grid, 1-D helper) as PatchEmbed calls it: np.meshgrid(grid_w, grid_h) puts
    the x coordinate in grid[0], and grid[0] feeds the FIRST half of the channels (published quirk, kept)."""
    gh = np.arange(grid_h, dtype=np.float32) / (grid_h / base_size) / interpolation_scale
    gw = np.arange(grid_w, dtype=np.float32) / (grid_w / base_size) / interpolation_scale
    grid = np.stack(np.meshgrid(gw, gh), axis=0).reshape(2, 1, grid_w, grid_h)

    def one(d, pos):
        omega = np.arange(d // 2, dtype=np.float64)
        omega /= d / 2.0
        omega = 1.0 / 10000 ** omega
        out = np.einsum("m,d->md", pos.reshape(-1), omega)
        return np.concatenate([np.sin(out), np.cos(out)], axis=1)

    emb = np.concatenate([one(embed_dim // 2, grid[0]), one(embed_dim // 2, grid[1])], axis=1)
    return torch.from_numpy(emb).float()


def attention(P, pfx, x, enc, heads, bias_mask, store, mid, want_map=False):
    """Attention + AttnProcessor2_0 with biases and an additive (B,1,K) mask (attention_processor.py:3244-3331); with
    want_map the eager AttnStoreProcessor (components/attention.py:176-263): softmax(q k^T * scale + mask) -> hook `map`."""
    src = x if enc is None else enc
    q = _lin(P, pfx + ".to_q", x); k = _lin(P, pfx + ".to_k", src); v = _lin(P, pfx + ".to_v", src)
    store.gather(mid, q, "q"); store.gather(mid, k, "k"); store.gather(mid, v, "v")          # :3291-3294
    b, s, c = q.shape
    d = c // heads
    sp = lambda t: t.view(b, -1, heads, d).transpose(1, 2)
    m = None if bias_mask is None else bias_mask[:, None]                                 # (B,1,1,K) broadcast over heads
    if want_map:
        sc = torch.matmul(sp(q), sp(k).transpose(-1, -2)) * d ** -0.5
        probs = torch.softmax(sc if m is None else sc + m, dim=-1)
        store.gather(mid, probs, "map")                                                   # components/attention.py:238-244
        o = torch.matmul(probs, sp(v))
        return _lin(P, pfx + ".to_out.0", o.transpose(1, 2).reshape(b, s, c))
    o = F.scaled_dot_product_attention(sp(q), sp(k), sp(v), attn_mask=m)
    return _lin(P, pfx + ".to_out.0", o.transpose(1, 2).reshape(b, s, c))


def block(P, i, arch, x, enc, tvec, enc_bias, store, want_map=False):
    """BasicTransformerBlock.forward, ada_norm_single (attention.py:498-592)."""
    b = f"transformer_blocks.{i}"; mid = f"vit-block{i}"
    heads = arch["num_attention_heads"]; B = x.shape[0]
    mod = P[b + ".scale_shift_table"][None] + tvec.reshape(B, 6, -1)                      # :498-503
    sh_a, sc_a, g_a, sh_m, sc_m, g_m = mod.chunk(6, dim=1)
    n = F.layer_norm(x, (x.shape[-1],), None, None, 1e-6) * (1 + sc_a) + sh_a
    x = g_a * attention(P, b + ".attn1", n, None, heads, None, store, mid + "-self", want_map) + x   # :514-526
    x = attention(P, b + ".attn2", x, enc, heads, enc_bias, store, mid + "-cross", want_map) + x     # :541-558 (no norm2, no gate)
    n = F.layer_norm(x, (x.shape[-1],), None, None, 1e-6) * (1 + sc_m) + sh_m             # :570-573
    h = F.gelu(_lin(P, b + ".ff.net.0.proj", n), approximate="tanh")
    store.gather(mid + "-ffn", h, "inner")                                                # :1255-1257
    x = g_m * _lin(P, b + ".ff.net.2", h) + x                                             # :583-586
    store.gather(mid, x, "out")                                                           # :589-590
    return x


def pixart_forward(P, arch, hidden_states, encoder_hidden_states, timestep, encoder_attention_mask=None, store=None,
                   want_map=None):
    """Transformer2DModel.forward, patched inputs + ada_norm_single (transformer_2d.py:404-475). Returns (B, out, H, W).
    want_map=None follows the reference: a requested '*map*' id (or accept-all) installs the eager processor everywhere."""
    store = store if store is not None else Store({"__none__": True})
    if want_map is None:
        want_map = store.accept_all or any("map" in k and v for k, v in store.to_store.items())
    C = inner_dim(arch); p = arch["patch_size"]
    B, _, H, W = hidden_states.shape
    enc_bias = None
    if encoder_attention_mask is not None:                                                # :397-399
        enc_bias = ((1 - encoder_attention_mask.float()) * -10000.0)[:, None]
    gh, gw = H // p, W // p
    x = F.conv2d(hidden_states.float(), P["pos_embed.proj.weight"], P["pos_embed.proj.bias"], stride=p)   # PatchEmbed
    x = x.flatten(2).transpose(1, 2)
    x = x + sincos_pos_embed(C, gh, gw, arch["sample_size"] // p, arch["interpolation_scale"])[None]
    emb = _lin(P, "adaln_single.emb.timestep_embedder.linear_2",
               F.silu(_lin(P, "adaln_single.emb.timestep_embedder.linear_1", timestep_sinusoid(timestep, 256))))
    tvec = _lin(P, "adaln_single.linear", F.silu(emb))                                    # AdaLayerNormSingle -> (B, 6C)
    enc = _lin(P, "caption_projection.linear_2",
               F.gelu(_lin(P, "caption_projection.linear_1", encoder_hidden_states.float()), approximate="tanh"))
    for i in range(arch["num_layers"]):
        x = block(P, i, arch, x, enc, tvec, enc_bias, store, want_map)
    shift, scale = (P["scale_shift_table"][None] + emb[:, None]).chunk(2, dim=1)          # :552-556
    x = F.layer_norm(x, (C,), None, None, 1e-6) * (1 + scale) + shift
    x = _lin(P, "proj_out", x)
    oc = arch["out_channels"]
    x = x.reshape(B, gh, gw, p, p, oc)                                                    # unpatchify :563-570
    return torch.einsum("nhwpqc->nchpwq", x).reshape(B, oc, gh * p, gw * p)


def hook_ids(arch, include_dropped=False, maps=False):
    ids = []
    for i in range(arch["num_layers"]):
        b = f"vit-block{i}"
        ids += [b + "-self-q", b + "-self-k", b + "-self-v"] + ([b + "-self-map"] if maps else []) + [b + "-cross-q"]
        if include_dropped:
            ids += [b + "-cross-k", b + "-cross-v"]
        ids += ([b + "-cross-map"] if maps else []) + [b + "-ffn-inner", b + "-out"]
    return ids


def flops_per_image(arch, n_img, n_txt):
    C = inner_dim(arch)
    per = 2.0 * n_img * C * 3 * C + 4.0 * n_img * n_img * C + 2.0 * n_img * C * C        # self attention
    per += 2.0 * n_img * C * C * 2 + 2.0 * n_txt * C * 2 * C + 4.0 * n_img * n_txt * C    # cross attention
    per += 2.0 * n_img * C * 4 * C * 2                                                    # feed forward
    io = 2.0 * n_txt * (arch["caption_channels"] * C + C * C)
    return arch["num_layers"] * per + io
