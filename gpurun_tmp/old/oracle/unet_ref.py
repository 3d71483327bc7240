"""CPU ORACLE (test infrastructure only) — pure-PyTorch fp32 restatement of the
reference's single-timestep UNet forward with per-layer activation hooks.

This file is a CHECKER.  Only tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg may import it; the product path
(generic-diffusion-feature_amd/) never does and fails loudly without its HIP
library.

Parity pinning: every block function below is checked against the reference's
own importable modules (tests/golden/gen_golden.py, run in the build container
where /root/reference exists) and against the committed golden vectors in
tests/golden/*.npz.  The un-vendored diffusers==0.32.2 pieces (GEGLU, Timesteps,
TimestepEmbedding, unet_2d_blocks wiring) are restated from the published
algorithm; their *structure* is pinned by the reference's ordered hook-id dumps
(feature/configs/config_15_full.json, config_xl_full.json).  No reference test
pins full-UNet numerics (the reference has no tests) => at whole-UNet level the
oracle is "parity unpinned" by the reference and pinned only block-wise.

Reference citations are relative to /root/reference/feature/.
"""
import math
from collections import OrderedDict

import torch
import torch.nn.functional as F

# --------------------------------------------------------------------------- #
# architecture descriptors (hyper-parameters of the public model configs the
# reference downloads: components/models.py:18-56)
# --------------------------------------------------------------------------- #
ARCHS = {
    # SD1.5: components/models.py:18-29 (runwayml/stable-diffusion-v1-5 config.json)
    "1-5": dict(
        in_channels=4, out_channels=4,
        block_out_channels=(320, 640, 1280, 1280),
        down_attn=(True, True, True, False),
        layers_per_block=2,
        transformer_layers=(1, 1, 1, 1),
        heads=(8, 8, 8, 8),
        cross_dim=768,
        linear_proj=False,
        addition_embed=None,
        time_embed_dim=1280,
    ),
    # SDXL / Playground-v2: components/models.py:43-70
    "xl": dict(
        in_channels=4, out_channels=4,
        block_out_channels=(320, 640, 1280),
        down_attn=(False, True, True),
        layers_per_block=2,
        transformer_layers=(1, 2, 10),
        heads=(5, 10, 20),
        cross_dim=2048,
        linear_proj=True,
        addition_embed="text_time",
        addition_time_embed_dim=256,
        add_in_dim=2816,
        time_embed_dim=1280,
    ),
}
ARCHS["pgv2"] = ARCHS["xl"]
# SD 2.1-base: components/models.py:30-42 (stabilityai/stable-diffusion-2-1-base config.json): SD1.5 topology,
# attention_head_dim (5,10,20,20) [= head counts], cross dim 1024, use_linear_projection
ARCHS["2-1"] = dict(ARCHS["1-5"], heads=(5, 10, 20, 20), cross_dim=1024, linear_proj=True)


def tiny_arch(base="xl", channels=None, heads=None, cross_dim=64, max_depth=2, time_embed_dim=256):
    """A shrunken architecture with the same topology (second-scale tests).
    Defaults: xl -> channels (64,128,256), dim_head 64; 1-5 -> channels (320,640,640,640), heads (8,8,4,4)
    (dim_head 40/80/160 like the real model)."""
    a = dict(ARCHS[base])
    if channels is None:
        channels = (320, 640, 640, 640) if base == "1-5" else (64, 128, 256, 256)[:len(a["block_out_channels"])]
    if heads is None:
        heads = tuple(max(1, c // 64) for c in channels) if base != "1-5" else (8, 8, 4, 4)[:len(channels)]
    a["block_out_channels"] = tuple(channels)
    a["heads"] = tuple(heads)
    a["transformer_layers"] = tuple(min(t, max_depth) for t in a["transformer_layers"])
    a["cross_dim"] = cross_dim
    a["time_embed_dim"] = time_embed_dim
    if a["addition_embed"]:
        a["addition_time_embed_dim"] = 32
        a["pooled_dim"] = 64
        a["add_in_dim"] = 64 + 6 * 32
    return a


# --------------------------------------------------------------------------- #
# hook sink: restatement of FeatureStore.store / FeatureGatherer.gather
# (components/feature_extractor.py:31-76, :83-89)
# --------------------------------------------------------------------------- #
class Store:
    def __init__(self, to_store=None, resize_ratio=1, out_dtype=torch.float16):
        self.to_store = to_store or {}
        self.accept_all = not bool(to_store)          # :10-15
        self.resize_ratio = resize_ratio
        self.out_dtype = out_dtype
        self.feats = OrderedDict()
        self.order = []                                # every gather() id, pre-filter

    def gather(self, module_id, feat, feat_id):
        fid = module_id + "-" + feat_id                # :88-89
        self.order.append(fid)
        if not (self.accept_all or self.to_store.get(fid, False)):   # :36
            return
        if "cross-k" in fid or "cross-v" in fid:       # :38-39
            return
        if feat.dim() == 3:                            # :46-48  b (h w) c -> b c h w
            b, n, c = feat.shape
            s = int(math.sqrt(n))
            feat = feat.reshape(b, s, n // s, c).permute(0, 3, 1, 2)
        if self.resize_ratio > 1:                      # :51-53
            tgt = (feat.shape[2] // self.resize_ratio, feat.shape[3] // self.resize_ratio)
            feat = F.adaptive_avg_pool2d(feat, tgt)
        feat = feat.clone()                            # :56 TF.normalize(mean=0,std=1) == clone
        if self.out_dtype is not None:                 # :59-60
            feat = feat.to(self.out_dtype)
        self.feats[fid] = feat.detach()                # :63-69


# --------------------------------------------------------------------------- #
# block restatements
# --------------------------------------------------------------------------- #
def timestep_sinusoid(t, dim, flip_sin_to_cos=True, shift=0.0, max_period=10000):
    """diffusers==0.32.2 embeddings.get_timestep_embedding (un-vendored; call site
    diffusers/models/unet/unet_2d_condition.py:910-934)."""
    half = dim // 2
    exponent = -math.log(max_period) * torch.arange(half, dtype=torch.float32) / (half - shift)
    emb = t.float()[:, None] * torch.exp(exponent)[None, :]
    emb = torch.cat([torch.sin(emb), torch.cos(emb)], dim=-1)
    if flip_sin_to_cos:
        emb = torch.cat([emb[:, half:], emb[:, :half]], dim=-1)
    return emb


def _stream(x):
    """Identity. Residual-stream write point (precision studies monkeypatch this)."""
    return x


def _lin(P, name, x):
    return F.linear(x, P[name + ".weight"], P.get(name + ".bias"))


def time_embed(P, arch, timestep, text_embeds=None, time_ids=None, act_dtype=None):
    """unet_2d_condition.py:1142-1162 (get_time_embed, time_embedding, get_aug_embed text_time :968-984)."""
    c0 = arch["block_out_channels"][0]
    t_emb = timestep_sinusoid(timestep, c0)
    if act_dtype is not None:                          # `t_emb.to(dtype=sample.dtype)` :933
        t_emb = t_emb.to(act_dtype).float()
    emb = _lin(P, "time_embedding.linear_2", F.silu(_lin(P, "time_embedding.linear_1", t_emb)))
    if arch["addition_embed"] == "text_time":
        b = text_embeds.shape[0]
        tid = timestep_sinusoid(time_ids.flatten(), arch["addition_time_embed_dim"])
        tid = tid.reshape(b, -1)
        add = torch.cat([text_embeds.float(), tid], dim=-1)
        if act_dtype is not None:                      # `add_embeds.to(emb.dtype)` :982
            add = add.to(act_dtype).float()
        aug = _lin(P, "add_embedding.linear_2", F.silu(_lin(P, "add_embedding.linear_1", add)))
        emb = emb + aug
    return emb


def resnet_block(P, pfx, x, emb, store, mid, eps=1e-5, groups=32):
    """ResnetBlock2D.forward — diffusers/models/resnet.py:320-379."""
    h = F.group_norm(x, groups, P[pfx + ".norm1.weight"], P[pfx + ".norm1.bias"], eps)
    h = F.silu(h)
    h = F.conv2d(h, P[pfx + ".conv1.weight"], P[pfx + ".conv1.bias"], padding=1)
    t = _lin(P, pfx + ".time_emb_proj", F.silu(emb))[:, :, None, None]      # :343-346
    h = h + t                                                               # :350
    h = F.group_norm(h, groups, P[pfx + ".norm2.weight"], P[pfx + ".norm2.bias"], eps)
    h = F.silu(h)
    h = F.conv2d(h, P[pfx + ".conv2.weight"], P[pfx + ".conv2.bias"], padding=1)
    if (pfx + ".conv_shortcut.weight") in P:                                # :368-369
        x = F.conv2d(x, P[pfx + ".conv_shortcut.weight"], P[pfx + ".conv_shortcut.bias"])
    store.gather(mid, h, "increment")                                       # :371-372
    out = _stream(x + h)                                                    # :374 (/1.0)
    store.gather(mid, out, "out")                                           # :376-377
    return out


def attention(P, pfx, x, ctx, heads, store, mid, want_map):
    """Attention + AttnProcessor2_0.__call__ (diffusers/models/attention_processor.py:3244-3331)
    or, when maps are requested, AttnStoreProcessor.__call__ (components/attention.py:176-263)."""
    enc = x if ctx is None else ctx
    q = F.linear(x, P[pfx + ".to_q.weight"])
    k = F.linear(enc, P[pfx + ".to_k.weight"])
    v = F.linear(enc, P[pfx + ".to_v.weight"])
    store.gather(mid, q, "q")                                               # :3291-3294
    store.gather(mid, k, "k")
    store.gather(mid, v, "v")
    b, s, c = q.shape
    d = c // heads
    qh = q.view(b, s, heads, d).transpose(1, 2)
    kh = k.view(b, -1, heads, d).transpose(1, 2)
    vh = v.view(b, -1, heads, d).transpose(1, 2)
    scale = d ** -0.5                                                       # attention_processor.py:166
    if want_map:
        # get_attention_scores: baddbmm(alpha=scale) -> softmax (attention_processor.py:640-685)
        probs = torch.softmax(torch.matmul(qh, kh.transpose(-1, -2)) * scale, dim=-1)
        store.gather(mid, probs, "map")                                     # components/attention.py:238-244
        o = torch.matmul(probs, vh)
    else:
        o = F.scaled_dot_product_attention(qh, kh, vh)                      # :3311-3313
    o = o.transpose(1, 2).reshape(b, s, c)
    return _lin(P, pfx + ".to_out.0", o)                                    # :3319


def feed_forward(P, pfx, x, store, mid):
    """FeedForward.forward (diffusers/models/attention.py:1249-1258) + GEGLU
    (diffusers==0.32.2 activations.GEGLU, un-vendored: proj -> chunk(2) -> h * gelu(gate), erf GELU)."""
    hg = _lin(P, pfx + ".net.0.proj", x)
    h, g = hg.chunk(2, dim=-1)
    inner = h * F.gelu(g)
    store.gather(mid, inner, "inner")                                       # :1255-1257
    return _lin(P, pfx + ".net.2", inner)


def basic_transformer_block(P, pfx, x, ctx, heads, store, mid, want_map):
    """BasicTransformerBlock.forward, norm_type == 'layer_norm' — diffusers/models/attention.py:469-592."""
    c = x.shape[-1]
    n = F.layer_norm(x, (c,), P[pfx + ".norm1.weight"], P[pfx + ".norm1.bias"], 1e-5)
    x = _stream(attention(P, pfx + ".attn1", n, None, heads, store, mid + "-self", want_map) + x)      # :514-526
    n = F.layer_norm(x, (c,), P[pfx + ".norm2.weight"], P[pfx + ".norm2.bias"], 1e-5)
    x = _stream(attention(P, pfx + ".attn2", n, ctx, heads, store, mid + "-cross", want_map) + x)      # :535-558
    n = F.layer_norm(x, (c,), P[pfx + ".norm3.weight"], P[pfx + ".norm3.bias"], 1e-5)
    x = _stream(feed_forward(P, pfx + ".ff", n, store, mid + "-ffn") + x)                               # :564-586
    store.gather(mid, x, "out")                                                                # :589-590
    return x


def transformer_2d(P, pfx, x, ctx, heads, depth, linear_proj, store, mid, want_map):
    """Transformer2DModel.forward, continuous input — diffusers/models/transformers/transformer_2d.py
    :404-407 (input), :482-495 (_operate_on_continuous_inputs), :417-450 (blocks), :517-530 (output), :474-475 hook."""
    b, c, hh, ww = x.shape
    res = x
    h = F.group_norm(x, 32, P[pfx + ".norm.weight"], P[pfx + ".norm.bias"], 1e-6)   # eps 1e-6 (:175-177)
    if linear_proj:
        h = h.permute(0, 2, 3, 1).reshape(b, hh * ww, c)
        h = _lin(P, pfx + ".proj_in", h)
    else:
        h = F.conv2d(h, P[pfx + ".proj_in.weight"], P[pfx + ".proj_in.bias"])
        h = h.permute(0, 2, 3, 1).reshape(b, hh * ww, c)
    for i in range(depth):
        h = basic_transformer_block(P, f"{pfx}.transformer_blocks.{i}", h, ctx, heads, store,
                                    f"{mid}-block{i}", want_map)
    if linear_proj:
        h = _lin(P, pfx + ".proj_out", h)
        h = h.reshape(b, hh, ww, c).permute(0, 3, 1, 2).contiguous()
    else:
        h = h.reshape(b, hh, ww, c).permute(0, 3, 1, 2).contiguous()
        h = F.conv2d(h, P[pfx + ".proj_out.weight"], P[pfx + ".proj_out.bias"])
    out = _stream(h + res)
    store.gather(mid, out, "out")
    return out


def downsample(P, pfx, x, store, mid):
    """Downsample2D.forward — diffusers/models/downsampling.py:132-152 (conv stride 2, padding 1)."""
    y = F.conv2d(x, P[pfx + ".conv.weight"], P[pfx + ".conv.bias"], stride=2, padding=1)
    store.gather(mid, y, "out")
    return y


def upsample(P, pfx, x, store, mid):
    """Upsample2D.forward — diffusers/models/upsampling.py:142-195 (nearest x2 then conv3x3)."""
    y = F.interpolate(x, scale_factor=2.0, mode="nearest")                  # :176-177
    y = F.conv2d(y, P[pfx + ".conv.weight"], P[pfx + ".conv.bias"], padding=1)
    store.gather(mid, y, "out")                                             # :192-193
    return y


# --------------------------------------------------------------------------- #
# whole UNet (wiring of diffusers==0.32.2 unet_2d_blocks, un-vendored; pinned by
# the ordered id dumps config_15_full.json / config_xl_full.json)
# --------------------------------------------------------------------------- #
def unet_forward(P, arch, sample, timestep, ctx, text_embeds=None, time_ids=None,
                 store=None, want_map=None, act_dtype=None):
    """UNet2DConditionModel.forward — diffusers/models/unet/unet_2d_condition.py:1040-1319.
    P: state dict (diffusers names), fp32 tensors.  Returns noise_pred (B,4,H,W)."""
    store = store or Store()
    if want_map is None:
        # diffusion_feature.py:72-77: any requested '*map*' id (or accept-all) swaps in the eager processor
        want_map = store.accept_all or any("map" in k and v for k, v in store.to_store.items())
    boc = arch["block_out_channels"]
    L = len(boc)
    nl = arch["layers_per_block"]
    sample = sample.float()
    ctx = ctx.float()
    if timestep.dim() == 0:
        timestep = timestep[None]
    timestep = timestep.expand(sample.shape[0])                              # :932
    emb = time_embed(P, arch, timestep, text_embeds, time_ids, act_dtype)

    store.gather("unet", sample, "in")                                       # :1169-1170
    h = F.conv2d(sample, P["conv_in.weight"], P["conv_in.bias"], padding=1)
    store.gather("unet", h, "after-conv-in")                                 # :1172-1173

    skips = [h]
    for lv in range(L):                                                      # :1212-1234
        for r in range(nl):
            h = resnet_block(P, f"down_blocks.{lv}.resnets.{r}", h, emb, store, f"down-level{lv}-repeat{r}-res")
            if arch["down_attn"][lv]:
                h = transformer_2d(P, f"down_blocks.{lv}.attentions.{r}", h, ctx, arch["heads"][lv],
                                   arch["transformer_layers"][lv], arch["linear_proj"], store,
                                   f"down-level{lv}-repeat{r}-vit", want_map)
            skips.append(h)
        if lv != L - 1:
            h = downsample(P, f"down_blocks.{lv}.downsamplers.0", h, store, f"down-level{lv}-downsampler")
            skips.append(h)

    # mid: resnet, attention, resnet (:1248-1259)
    h = resnet_block(P, "mid_block.resnets.0", h, emb, store, "mid-repeat0-res")
    h = transformer_2d(P, "mid_block.attentions.0", h, ctx, arch["heads"][-1], arch["transformer_layers"][-1],
                       arch["linear_proj"], store, "mid-vit", want_map)
    h = resnet_block(P, "mid_block.resnets.1", h, emb, store, "mid-repeat1-res")

    for i in range(L):                                                       # :1273-1301
        lv = L - 1 - i                                                       # resolution level of this up block
        for r in range(nl + 1):
            h = torch.cat([h, skips.pop()], dim=1)
            h = resnet_block(P, f"up_blocks.{i}.resnets.{r}", h, emb, store, f"up-level{i}-repeat{r}-res")
            if arch["down_attn"][lv]:
                h = transformer_2d(P, f"up_blocks.{i}.attentions.{r}", h, ctx, arch["heads"][lv],
                                   arch["transformer_layers"][lv], arch["linear_proj"], store,
                                   f"up-level{i}-repeat{r}-vit", want_map)
        if i != L - 1:
            h = upsample(P, f"up_blocks.{i}.upsamplers.0", h, store, f"up-level{i}-upsampler")

    h = F.group_norm(h, 32, P["conv_norm_out.weight"], P["conv_norm_out.bias"], 1e-5)   # :1304-1307
    h = F.silu(h)
    h = F.conv2d(h, P["conv_out.weight"], P["conv_out.bias"], padding=1)
    store.gather("unet", h, "out")                                           # :1309-1310
    return h


# --------------------------------------------------------------------------- #
# parameter inventory (diffusers state-dict names and shapes) + synthetic init
# --------------------------------------------------------------------------- #
def param_shapes(arch):
    """OrderedDict name -> shape, in diffusers' UNet2DConditionModel state_dict naming."""
    S = OrderedDict()
    boc = arch["block_out_channels"]
    L = len(boc)
    nl = arch["layers_per_block"]
    te = arch["time_embed_dim"]
    cd = arch["cross_dim"]

    def conv(n, co, ci, k):
        S[n + ".weight"] = (co, ci, k, k)
        S[n + ".bias"] = (co,)

    def lin(n, co, ci, bias=True):
        S[n + ".weight"] = (co, ci)
        if bias:
            S[n + ".bias"] = (co,)

    def norm(n, c):
        S[n + ".weight"] = (c,)
        S[n + ".bias"] = (c,)

    def resnet(p, ci, co):
        norm(p + ".norm1", ci); conv(p + ".conv1", co, ci, 3); lin(p + ".time_emb_proj", co, te)
        norm(p + ".norm2", co); conv(p + ".conv2", co, co, 3)
        if ci != co:
            conv(p + ".conv_shortcut", co, ci, 1)

    def vit(p, c, depth):
        norm(p + ".norm", c)
        if arch["linear_proj"]:
            lin(p + ".proj_in", c, c)
        else:
            conv(p + ".proj_in", c, c, 1)
        for i in range(depth):
            b = f"{p}.transformer_blocks.{i}"
            norm(b + ".norm1", c)
            for n_ in ("to_q", "to_k", "to_v"):
                lin(f"{b}.attn1.{n_}", c, c, bias=False)
            lin(b + ".attn1.to_out.0", c, c)
            norm(b + ".norm2", c)
            lin(b + ".attn2.to_q", c, c, bias=False)
            lin(b + ".attn2.to_k", c, cd, bias=False)
            lin(b + ".attn2.to_v", c, cd, bias=False)
            lin(b + ".attn2.to_out.0", c, c)
            norm(b + ".norm3", c)
            lin(b + ".ff.net.0.proj", 8 * c, c)
            lin(b + ".ff.net.2", c, 4 * c)
        if arch["linear_proj"]:
            lin(p + ".proj_out", c, c)
        else:
            conv(p + ".proj_out", c, c, 1)

    conv("conv_in", boc[0], arch["in_channels"], 3)
    lin("time_embedding.linear_1", te, boc[0]); lin("time_embedding.linear_2", te, te)
    if arch["addition_embed"] == "text_time":
        lin("add_embedding.linear_1", te, arch["add_in_dim"]); lin("add_embedding.linear_2", te, te)
    ci = boc[0]
    for lv in range(L):
        co = boc[lv]
        for r in range(nl):
            resnet(f"down_blocks.{lv}.resnets.{r}", ci, co)
            if arch["down_attn"][lv]:
                vit(f"down_blocks.{lv}.attentions.{r}", co, arch["transformer_layers"][lv])
            ci = co
        if lv != L - 1:
            conv(f"down_blocks.{lv}.downsamplers.0.conv", co, co, 3)
    cm = boc[-1]
    resnet("mid_block.resnets.0", cm, cm)
    vit("mid_block.attentions.0", cm, arch["transformer_layers"][-1])
    resnet("mid_block.resnets.1", cm, cm)
    rev = list(reversed(boc))
    prev = rev[0]
    for i in range(L):
        co = rev[i]
        cin_skip = rev[min(i + 1, L - 1)]
        lv = L - 1 - i
        for r in range(nl + 1):
            skip_c = cin_skip if r == nl else co
            in_c = prev if r == 0 else co
            resnet(f"up_blocks.{i}.resnets.{r}", in_c + skip_c, co)
            if arch["down_attn"][lv]:
                vit(f"up_blocks.{i}.attentions.{r}", co, arch["transformer_layers"][lv])
        if i != L - 1:
            conv(f"up_blocks.{i}.upsamplers.0.conv", co, co, 3)
        prev = co
    norm("conv_norm_out", boc[0])
    conv("conv_out", arch["out_channels"], boc[0], 3)
    return S


def synth_params(arch, seed=0, dtype=torch.float32):
    """Seeded synthetic weights (no checkpoints exist offline): W ~ N(0, 1/fan_in), bias ~ 0.05 N,
    norm gamma = 1 + 0.1 N, beta = 0.1 N; all rounded to fp16 then upcast to `dtype`."""
    g = torch.Generator().manual_seed(seed)
    P = OrderedDict()
    for name, shape in param_shapes(arch).items():
        is_norm = ".norm" in name or name.startswith("conv_norm_out")
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


def synth_params_heavy(arch, seed=0, dtype=torch.float32, sigma=0.5, n_outlier=3, outlier_gain=32.0):
    """Heavy-tailed synthetic weights: what real SD checkpoints look like and N(0, 1/fan_in) does not.

    * every matrix / conv gets LOG-NORMAL per-output-channel scales exp(sigma N): rows of very different magnitude;
    * `n_outlier` fixed channels per width are OUTLIER channels of the residual stream: every layer that writes the stream (conv_in,
      resnet conv2, attn to_out.0, ff.net.2, proj_out: the additive branches, whose inputs are normalised) has those output rows multiplied by
      `outlier_gain`, so the stream carries a few channels ~32x the others through the whole depth (the massive activations of trained
      diffusion / transformer checkpoints); the gammas of the norms that READ the stream are 1/8 on those channels, as trained
      models compensate;
    * all norm gammas log-normal (sigma 0.3).
    Same rounding contract as synth_params: every value is fp16-representable."""
    g = torch.Generator().manual_seed(seed + 7919)
    P = synth_params(arch, seed=seed, dtype=torch.float32)
    widths = set(arch["block_out_channels"])
    idx = {c: torch.randperm(c, generator=torch.Generator().manual_seed(1000 + c))[:n_outlier] for c in widths}
    # (the 1x1 shortcuts and the sampler convs READ the raw stream: its outlier channels pass through them without a gain of their own)
    writers = (".conv2.weight", ".to_out.0.weight", ".ff.net.2.weight", ".proj_out.weight")
    for name, w in list(P.items()):
        if not name.endswith(".weight"):
            continue
        is_norm = ".norm" in name or name.startswith("conv_norm_out")
        if not is_norm:
            sc = torch.exp(sigma * torch.randn(w.shape[0], generator=g))
            if (name == "conv_in.weight" or name.endswith(writers)) and w.shape[0] in widths:
                sc[idx[w.shape[0]]] *= outlier_gain
            w = w * sc.view(-1, *([1] * (w.dim() - 1)))
        else:
            w = w * torch.exp(0.3 * torch.randn(w.shape[0], generator=g))
            reads_stream = not name.endswith(".norm2.weight") or "transformer_blocks" in name     # (resnet norm2 reads the conv1 output)
            if reads_stream and w.shape[0] in widths:
                w = w.clone()
                w[idx[w.shape[0]]] *= 0.125
        P[name] = w.half().to(dtype)
    return P


def synth_inputs(arch, batch, lat, seed=1, n_ctx=77, same_prompt=True):
    """Seeded inputs: latents N(0,1) fp16-rounded, ctx N(0,1), SDXL pooled + time_ids (diffusion_feature.py:324-337)."""
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(batch, arch["in_channels"], lat, lat, generator=g).half().float()
    ctx = torch.randn(1 if same_prompt else batch, n_ctx, arch["cross_dim"], generator=g).half().float()
    ctx = ctx.expand(batch, -1, -1).contiguous()
    out = dict(sample=x, ctx=ctx, timestep=torch.tensor([100.0]))
    if arch["addition_embed"] == "text_time":
        pd = arch.get("pooled_dim", 1280)
        pooled = torch.randn(1 if same_prompt else batch, pd, generator=g).half().float()
        out["text_embeds"] = pooled.expand(batch, -1).contiguous()
        img = float(lat * 8)
        out["time_ids"] = torch.tensor([[img, img, 0.0, 0.0, img, img]]).repeat(batch, 1)
    return out


def hook_ids(arch):
    """Every gather() id in execution order incl. the ones FeatureStore drops (cross-k/v)."""
    ids = ["unet-in", "unet-after-conv-in"]
    boc = arch["block_out_channels"]; L = len(boc); nl = arch["layers_per_block"]

    def res(m):
        ids.extend([m + "-res-increment", m + "-res-out"])

    def vit(m, depth):
        for i in range(depth):
            b = f"{m}-vit-block{i}"
            ids.extend([b + "-self-q", b + "-self-k", b + "-self-v", b + "-self-map",
                        b + "-cross-q", b + "-cross-k", b + "-cross-v", b + "-cross-map",
                        b + "-ffn-inner", b + "-out"])
        ids.append(m + "-vit-out")

    for lv in range(L):
        for r in range(nl):
            res(f"down-level{lv}-repeat{r}")
            if arch["down_attn"][lv]:
                vit(f"down-level{lv}-repeat{r}", arch["transformer_layers"][lv])
        if lv != L - 1:
            ids.append(f"down-level{lv}-downsampler-out")
    res("mid-repeat0"); vit("mid", arch["transformer_layers"][-1]); res("mid-repeat1")
    for i in range(L):
        lv = L - 1 - i
        for r in range(nl + 1):
            res(f"up-level{i}-repeat{r}")
            if arch["down_attn"][lv]:
                vit(f"up-level{i}-repeat{r}", arch["transformer_layers"][lv])
        if i != L - 1:
            ids.append(f"up-level{i}-upsampler-out")
    ids.append("unet-out")
    return ids


def stored_hook_ids(arch):
    """What an accept-all FeatureStore keeps (cross-k / cross-v dropped) == the reference's *_full.json key order."""
    return [i for i in hook_ids(arch) if not (i.endswith("cross-k") or i.endswith("cross-v"))]
