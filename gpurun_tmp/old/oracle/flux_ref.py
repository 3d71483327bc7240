"""CPU ORACLE (test infrastructure only) — pure-PyTorch fp32 restatement of the reference's
Flux MMDiT single forward with its per-layer activation hooks.

This file is a CHECKER.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
import it; the product path (generic-diffusion-feature_amd/) never does.

What is restated, and from where (paths relative to /root/reference/feature/):
  FluxTransformer2DModel.forward        diffusers/models/transformers/transformer_flux.py:414-603
  FluxTransformerBlock.forward          transformer_flux.py:167-226   (hooks `norm-out`, `out` — BOTH store
                                        norm_hidden_states, :200-211, reference quirk kept)
  FluxSingleTransformerBlock.forward    transformer_flux.py:86-112    (hook `out` on [:, text_len:])
  FluxAttnProcessor2_0.__call__         diffusers/models/attention_processor.py:2266-2362 (hooks q/k/v pre-norm,
                                        pre-RoPE, image tokens only; `attn-out`)
  FeedForward.forward                   diffusers/models/attention.py:1249-1258 (hook `ffn-inner`, `ff` only)
  hook ids                              components/feature_extractor.py:98-123
Un-vendored diffusers==0.32.2 pieces restated from the published algorithm (the reference imports them at
transformer_flux.py:34-38, attention_processor.py:141,2331, attention.py:22):
  AdaLayerNormZero / AdaLayerNormZeroSingle / AdaLayerNormContinuous, RMSNorm, FluxPosEmbed +
  get_1d_rotary_pos_embed, apply_rotary_emb, CombinedTimestep(Guidance)TextProjEmbeddings, Timesteps,
  TimestepEmbedding, PixArtAlphaTextProjection, activations.GELU(approximate="tanh").

Parity pinning: tests/golden/gen_golden_flux.py runs the reference's OWN transformer_flux.py /
attention_processor.py / attention.py / feature_extractor.py modules (imported from /root/reference by
oracle/ref_blocks.py, with the un-vendored classes above supplied as scaffolding) on a tiny MMDiT and commits
inputs + outputs + every hook as tests/golden/flux_tiny.npz; tests/test_oracle_golden.py checks this file against
it.  The un-vendored pieces themselves have no reference-side vector => for them "parity unpinned" (restated
from the published diffusers algorithm), exactly as for the UNet's un-vendored wiring.
"""
import math
from collections import OrderedDict

import torch
import torch.nn.functional as F

from .unet_ref import Store, timestep_sinusoid  # noqa: F401  (same FeatureStore restatement, same Timesteps)

# FLUX.1-dev transformer/config.json (the reference downloads it: components/models.py:150-169)
ARCH_FLUX_DEV = dict(in_channels=64, num_layers=19, num_single_layers=38, attention_head_dim=128,
                     num_attention_heads=24, joint_attention_dim=4096, pooled_projection_dim=768,
                     guidance_embeds=True, axes_dims_rope=(16, 56, 56), mlp_ratio=4)


def tiny_arch(heads=2, num_layers=2, num_single_layers=2, joint_dim=128, pooled_dim=64, guidance=True):
    """Same topology, shrunken widths (head dim stays 128: the RoPE axes (16,56,56) sum to it)."""
    a = dict(ARCH_FLUX_DEV)
    a.update(num_attention_heads=heads, num_layers=num_layers, num_single_layers=num_single_layers,
             joint_attention_dim=joint_dim, pooled_projection_dim=pooled_dim, guidance_embeds=guidance)
    return a


def inner_dim(arch):
    return arch["num_attention_heads"] * arch["attention_head_dim"]


def param_shapes(arch):
    """diffusers `state_dict()` names and shapes of FluxTransformer2DModel (transformer_flux.py:259-305)."""
    C = inner_dim(arch); D = arch["attention_head_dim"]; hid = int(C * arch["mlp_ratio"])
    S = OrderedDict()

    def lin(n, o, i):
        S[n + ".weight"] = (o, i); S[n + ".bias"] = (o,)

    lin("x_embedder", C, arch["in_channels"])
    lin("context_embedder", C, arch["joint_attention_dim"])
    lin("time_text_embed.timestep_embedder.linear_1", C, 256)
    lin("time_text_embed.timestep_embedder.linear_2", C, C)
    if arch["guidance_embeds"]:
        lin("time_text_embed.guidance_embedder.linear_1", C, 256)
        lin("time_text_embed.guidance_embedder.linear_2", C, C)
    lin("time_text_embed.text_embedder.linear_1", C, arch["pooled_projection_dim"])
    lin("time_text_embed.text_embedder.linear_2", C, C)
    for i in range(arch["num_layers"]):
        p = f"transformer_blocks.{i}"
        lin(p + ".norm1.linear", 6 * C, C)
        lin(p + ".norm1_context.linear", 6 * C, C)
        for n in ("to_q", "to_k", "to_v", "add_q_proj", "add_k_proj", "add_v_proj", "to_out.0", "to_add_out"):
            lin(p + ".attn." + n, C, C)
        for n in ("norm_q", "norm_k", "norm_added_q", "norm_added_k"):
            S[p + ".attn." + n + ".weight"] = (D,)
        lin(p + ".ff.net.0.proj", hid, C); lin(p + ".ff.net.2", C, hid)
        lin(p + ".ff_context.net.0.proj", hid, C); lin(p + ".ff_context.net.2", C, hid)
    for i in range(arch["num_single_layers"]):
        p = f"single_transformer_blocks.{i}"
        lin(p + ".norm.linear", 3 * C, C)
        lin(p + ".proj_mlp", hid, C)
        lin(p + ".proj_out", C, C + hid)
        for n in ("to_q", "to_k", "to_v"):
            lin(p + ".attn." + n, C, C)
        for n in ("norm_q", "norm_k"):
            S[p + ".attn." + n + ".weight"] = (D,)
    lin("norm_out.linear", 2 * C, C)
    lin("proj_out", arch["in_channels"], C)
    return S


def synth_params(arch, seed=0, dtype=torch.float32):
    """Seeded synthetic weights: W ~ N(0, 1/fan_in), bias ~ 0.05 N, RMSNorm gains 1 + 0.1 N; fp16-rounded."""
    g = torch.Generator().manual_seed(seed)
    P = OrderedDict()
    for name, shape in param_shapes(arch).items():
        if ".attn.norm_" in name:
            w = 1.0 + 0.1 * torch.randn(shape, generator=g)
        elif name.endswith(".weight"):
            w = torch.randn(shape, generator=g) / math.sqrt(shape[1])
        else:
            w = 0.05 * torch.randn(shape, generator=g)
        P[name] = w.half().to(dtype)
    return P


def latent_image_ids(h, w):
    """FluxImg2ImgPipeline._prepare_latent_image_ids (un-vendored pipeline; the reference calls the whole pipe,
    diffusion_feature.py:246-254): (h*w, 3) rows [0, y, x] over the PACKED latent grid (h = H/16, w = W/16)."""
    ids = torch.zeros(h, w, 3)
    ids[..., 1] = torch.arange(h)[:, None]
    ids[..., 2] = torch.arange(w)[None, :]
    return ids.reshape(h * w, 3)


def synth_inputs(arch, batch, grid, n_txt, seed=1, same_prompt=True):
    """hidden_states (B, grid*grid, in_channels) packed latents, encoder_hidden_states (B, n_txt, joint_dim),
    pooled (B, pooled_dim), timestep = t/1000 (the pipeline passes sigma-scaled t/1000), guidance scale 1
    (reference: `guidance_scale=1`, diffusion_feature.py:252), img_ids / txt_ids as the pipeline prepares them."""
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(batch, grid * grid, arch["in_channels"], generator=g).half().float()
    nb = 1 if same_prompt else batch
    ctx = torch.randn(nb, n_txt, arch["joint_attention_dim"], generator=g).half().float().expand(batch, -1, -1).contiguous()
    pooled = torch.randn(nb, arch["pooled_projection_dim"], generator=g).half().float().expand(batch, -1).contiguous()
    return dict(hidden_states=x, encoder_hidden_states=ctx, pooled_projections=pooled,
                timestep=torch.full((batch,), 0.1), guidance=torch.full((batch,), 1.0) if arch["guidance_embeds"] else None,
                img_ids=latent_image_ids(grid, grid), txt_ids=torch.zeros(n_txt, 3))


# --------------------------------------------------------------------------- #
# un-vendored diffusers==0.32.2 pieces (published algorithm)
# --------------------------------------------------------------------------- #
def _lin(P, name, x):
    return F.linear(x, P[name + ".weight"], P.get(name + ".bias"))


def rope_freqs(ids, axes_dim, theta=10000.0):
    """embeddings.FluxPosEmbed.forward + get_1d_rotary_pos_embed(use_real=True, repeat_interleave_real=True,
    freqs_dtype=float64): cos/sin tables (S, sum(axes_dim)) in fp32."""
    cos_out, sin_out = [], []
    pos = ids.float()
    for i, d in enumerate(axes_dim):
        freqs = 1.0 / (theta ** (torch.arange(0, d, 2, dtype=torch.float64)[: d // 2] / d))
        ang = torch.outer(pos[:, i].to(torch.float64), freqs)
        cos_out.append(ang.cos().repeat_interleave(2, dim=1).float())
        sin_out.append(ang.sin().repeat_interleave(2, dim=1).float())
    return torch.cat(cos_out, dim=-1), torch.cat(sin_out, dim=-1)


def apply_rope(x, cos, sin):
    """embeddings.apply_rotary_emb(use_real=True, use_real_unbind_dim=-1); x (B, heads, S, D)."""
    xr, xi = x.reshape(*x.shape[:-1], -1, 2).unbind(-1)
    rot = torch.stack([-xi, xr], dim=-1).flatten(3)
    return x * cos[None, None] + rot * sin[None, None]


def rms_norm(x, w, eps=1e-6):
    """normalization.RMSNorm.forward over the last dim (per head, dim_head)."""
    var = x.pow(2).mean(-1, keepdim=True)
    return x * torch.rsqrt(var + eps) * w


def layer_norm(x, eps=1e-6):
    return F.layer_norm(x, (x.shape[-1],), None, None, eps)


def time_text_embed(P, arch, timestep, guidance, pooled):
    """embeddings.CombinedTimestepGuidanceTextProjEmbeddings / CombinedTimestepTextProjEmbeddings
    (Timesteps(256, flip_sin_to_cos=True, shift=0) -> TimestepEmbedding; PixArtAlphaTextProjection(act silu))."""
    pfx = "time_text_embed."
    t = _lin(P, pfx + "timestep_embedder.linear_2", F.silu(_lin(P, pfx + "timestep_embedder.linear_1",
                                                                 timestep_sinusoid(timestep, 256))))
    if arch["guidance_embeds"]:
        t = t + _lin(P, pfx + "guidance_embedder.linear_2", F.silu(_lin(P, pfx + "guidance_embedder.linear_1",
                                                                         timestep_sinusoid(guidance, 256))))
    p = _lin(P, pfx + "text_embedder.linear_2", F.silu(_lin(P, pfx + "text_embedder.linear_1", pooled)))
    return t + p


# --------------------------------------------------------------------------- #
# in-tree blocks
# --------------------------------------------------------------------------- #
def flux_attention(P, pfx, arch, x, enc, cos, sin, store, mid, text_len, want_map=False):
    """FluxAttnProcessor2_0.__call__ (attention_processor.py:2266-2362); with want_map the eager FluxAttnStoreProcessor
    (components/attention.py:404-527): softmax(q k^T / sqrt(d)) materialised, hooks `cross-map` = probs[:, :, T:, :T] and
    `self-map` = probs[:, :, T:, T:] (image queries only), in that order, before the output projections (:493-502)."""
    heads = arch["num_attention_heads"]
    b = x.shape[0]
    q = _lin(P, pfx + ".to_q", x); k = _lin(P, pfx + ".to_k", x); v = _lin(P, pfx + ".to_v", x)
    if enc is not None:                                                   # :2283-2286
        store.gather(mid, q, "q"); store.gather(mid, k, "k"); store.gather(mid, v, "v")
    else:                                                                 # :2287-2291 image tokens only
        store.gather(mid, q[:, text_len:], "q"); store.gather(mid, k[:, text_len:], "k")
        store.gather(mid, v[:, text_len:], "v")
    d = q.shape[-1] // heads
    split = lambda t: t.view(b, -1, heads, d).transpose(1, 2)
    q, k, v = split(q), split(k), split(v)
    q = rms_norm(q, P[pfx + ".norm_q.weight"]); k = rms_norm(k, P[pfx + ".norm_k.weight"])   # :2300-2303
    if enc is not None:
        eq = split(_lin(P, pfx + ".add_q_proj", enc)); ek = split(_lin(P, pfx + ".add_k_proj", enc))
        ev = split(_lin(P, pfx + ".add_v_proj", enc))
        eq = rms_norm(eq, P[pfx + ".norm_added_q.weight"]); ek = rms_norm(ek, P[pfx + ".norm_added_k.weight"])
        q = torch.cat([eq, q], dim=2); k = torch.cat([ek, k], dim=2); v = torch.cat([ev, v], dim=2)   # :2327-2329
    q = apply_rope(q, cos, sin); k = apply_rope(k, cos, sin)             # :2331-2335
    if want_map:
        tl = enc.shape[1] if enc is not None else text_len
        probs = torch.softmax(torch.matmul(q, k.transpose(-1, -2)) / math.sqrt(d), dim=-1)   # components/attention.py:265-292
        store.gather(mid, probs[:, :, tl:, :tl], "cross-map")
        store.gather(mid, probs[:, :, tl:, tl:], "self-map")
        o = torch.matmul(probs, v)
    else:
        o = F.scaled_dot_product_attention(q, k, v)                       # :2337-2339
    o = o.transpose(1, 2).reshape(b, -1, heads * d)
    if enc is not None:
        n_enc = enc.shape[1]
        eo, o = o[:, :n_enc], o[:, n_enc:]
        o = _lin(P, pfx + ".to_out.0", o)
        eo = _lin(P, pfx + ".to_add_out", eo)
        store.gather(mid, o, "attn-out")                                  # :2355-2356
        return o, eo
    store.gather(mid, o[:, text_len:], "attn-out")                        # :2360-2361
    return o


def feed_forward(P, pfx, x, store=None, mid=None):
    """FeedForward(activation_fn='gelu-approximate') (attention.py:1228-1258): GELU(tanh) proj, hook, Linear."""
    h = F.gelu(_lin(P, pfx + ".net.0.proj", x), approximate="tanh")
    if store is not None:
        store.gather(mid, h, "inner")                                     # :1255-1257 (id prefix `...-ffn`)
    return _lin(P, pfx + ".net.2", h)


def double_block(P, i, arch, x, enc, temb, cos, sin, store, want_map=False):
    """FluxTransformerBlock.forward (transformer_flux.py:167-226)."""
    p = f"transformer_blocks.{i}"; mid = f"vit-block{i}"
    mod = _lin(P, p + ".norm1.linear", F.silu(temb))                      # AdaLayerNormZero
    sh_a, sc_a, g_a, sh_m, sc_m, g_m = mod.chunk(6, dim=1)
    nx = layer_norm(x) * (1 + sc_a[:, None]) + sh_a[:, None]
    cmod = _lin(P, p + ".norm1_context.linear", F.silu(temb))
    csh_a, csc_a, cg_a, csh_m, csc_m, cg_m = cmod.chunk(6, dim=1)
    ne = layer_norm(enc) * (1 + csc_a[:, None]) + csh_a[:, None]
    ao, eo = flux_attention(P, p + ".attn", arch, nx, ne, cos, sin, store, mid, None, want_map)
    x = x + g_a[:, None] * ao                                             # :191-192
    nx = layer_norm(x) * (1 + sc_m[:, None]) + sh_m[:, None]             # :194-195
    store.gather(mid, nx, "norm-out")                                     # :196-197
    ff = feed_forward(P, p + ".ff", nx, store, mid + "-ffn")
    x = x + g_m[:, None] * ff                                             # :199-202
    store.gather(mid, nx, "out")                                          # :206-207 (norm_hidden_states again)
    enc = enc + cg_a[:, None] * eo                                        # :211-212
    ne = layer_norm(enc) * (1 + csc_m[:, None]) + csh_m[:, None]
    enc = enc + cg_m[:, None] * feed_forward(P, p + ".ff_context", ne)    # :217-218
    return enc, x


def single_block(P, i, idx, arch, x, temb, cos, sin, store, text_len, want_map=False):
    """FluxSingleTransformerBlock.forward (transformer_flux.py:86-112); `idx` continues the double-block numbering
    (components/feature_extractor.py:112-122)."""
    p = f"single_transformer_blocks.{i}"; mid = f"vit-block{idx}"
    mod = _lin(P, p + ".norm.linear", F.silu(temb))                       # AdaLayerNormZeroSingle
    sh, sc, gate = mod.chunk(3, dim=1)
    nx = layer_norm(x) * (1 + sc[:, None]) + sh[:, None]
    mlp = F.gelu(_lin(P, p + ".proj_mlp", nx), approximate="tanh")       # :95
    ao = flux_attention(P, p + ".attn", arch, nx, None, cos, sin, store, mid, text_len, want_map)
    h = gate[:, None] * _lin(P, p + ".proj_out", torch.cat([ao, mlp], dim=2))   # :103-105
    x = x + h
    store.gather(mid, x[:, text_len:], "out")                             # :107-108
    return x


def flux_forward(P, arch, hidden_states, encoder_hidden_states, pooled_projections, timestep, img_ids, txt_ids,
                 guidance=None, store=None, want_map=None):
    """FluxTransformer2DModel.forward (transformer_flux.py:414-603). Returns (B, S, in_channels).
    want_map=None follows the reference: any requested '*map*' id (or accept-all) swaps in the eager processor
    (diffusion_feature.py:72-77)."""
    store = store if store is not None else Store({"__none__": True})
    if want_map is None:
        want_map = store.accept_all or any("map" in k and v for k, v in store.to_store.items())
    x = _lin(P, "x_embedder", hidden_states)                              # :470
    t = timestep.float() * 1000                                           # :472
    g = guidance.float() * 1000 if guidance is not None else None
    temb = time_text_embed(P, arch, t, g, pooled_projections)             # :478-482
    enc = _lin(P, "context_embedder", encoder_hidden_states)              # :483
    cos, sin = rope_freqs(torch.cat([txt_ids, img_ids], dim=0), arch["axes_dims_rope"])   # :498-499
    for i in range(arch["num_layers"]):
        enc, x = double_block(P, i, arch, x, enc, temb, cos, sin, store, want_map)
    text_len = enc.shape[1]
    x = torch.cat([enc, x], dim=1)                                        # :549
    for i in range(arch["num_single_layers"]):
        x = single_block(P, i, arch["num_layers"] + i, arch, x, temb, cos, sin, store, text_len, want_map)
    x = x[:, text_len:]                                                   # :591
    mod = _lin(P, "norm_out.linear", F.silu(temb))                        # AdaLayerNormContinuous: scale, shift
    scale, shift = mod.chunk(2, dim=1)
    x = layer_norm(x) * (1 + scale)[:, None] + shift[:, None]
    return _lin(P, "proj_out", x)                                         # :594


def hook_ids(arch, maps=False):
    """Every gather() id in execution order; maps=True adds the eager processor's `cross-map` / `self-map`
    (what an accept-all FeatureStore keeps: the reference then installs FluxAttnStoreProcessor, diffusion_feature.py:72-77)."""
    ids = []
    mp = lambda b: [b + "-cross-map", b + "-self-map"] if maps else []
    for i in range(arch["num_layers"]):
        b = f"vit-block{i}"
        ids += [b + "-q", b + "-k", b + "-v"] + mp(b) + [b + "-attn-out", b + "-norm-out", b + "-ffn-inner", b + "-out"]
    for j in range(arch["num_single_layers"]):
        b = f"vit-block{arch['num_layers'] + j}"
        ids += [b + "-q", b + "-k", b + "-v"] + mp(b) + [b + "-attn-out", b + "-out"]
    return ids


def flops_per_image(arch, n_img, n_txt):
    """2*MACs of every linear + QK^T + PV (74.4 TFLOP/img for FLUX.1-dev at 4096+512 tokens)."""
    C = inner_dim(arch); hid = int(C * arch["mlp_ratio"]); S = n_img + n_txt
    attn = 4.0 * S * S * C
    dbl = 2.0 * S * C * (3 * C) + attn + 2.0 * S * C * C + 2.0 * S * (2 * C * hid)
    sgl = 2.0 * S * C * (3 * C + hid) + attn + 2.0 * S * (C + hid) * C
    io = 2.0 * n_img * arch["in_channels"] * C * 2 + 2.0 * n_txt * arch["joint_attention_dim"] * C
    return arch["num_layers"] * dbl + arch["num_single_layers"] * sgl + io
