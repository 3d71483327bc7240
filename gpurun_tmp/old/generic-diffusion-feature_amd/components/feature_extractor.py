"""Hook sink + hook registry of the native path.

Mirrors the interface of the reference's feature/components/feature_extractor.py
(FeatureStore :8-80, FeatureGatherer :83-89, prepare_feature_extractor :92-288) so callers that read
`feature_store.to_store`, `.stored_feats`, `.reset()`, `.pause()/.resume()`, `.store_idx` keep working.
The difference: hooks are not Python callbacks inside the model.  The layer-id JSON is handed to
libgdf.so, whose kernels write the selected activations straight into caller-owned fp16 buffers;
`store()` only applies the reference's post-processing that is not already done in HBM.
"""
import json
import math

import torch
import torch.nn.functional as F


class FeatureStore:
    def __init__(self, to_store, resize_ratio, train_unet):
        if to_store:
            self.to_store = to_store
            self.accept_all = False
        else:                                   # None / {}: accept every layer (reference :9-15)
            self.to_store = {}
            self.accept_all = True
        self.feats = {}
        self.status = 'active'
        self.resize_ratio = resize_ratio
        self.train_unet = train_unet
        self.store_idx = None

    def pause(self):
        self.status = 'pause'

    def resume(self):
        self.status = 'active'

    def reset(self):
        self.feats = {}                         # a fresh dict: previously returned dicts stay valid

    def store(self, feat, feat_id):
        """Same filtering / post-processing order as the reference's store() (:31-76)."""
        if self.status == 'pause':
            return
        if not ((feat_id in self.to_store and self.to_store[feat_id]) or self.accept_all):
            return
        if 'cross-k' in feat_id or 'cross-v' in feat_id:     # 77 text tokens are not a square (:38-39)
            return
        if feat.dim() == 3:                                   # (B, HW, C) -> (B, C, H, W) view (:46-48)
            b, n, c = feat.shape
            s = int(math.sqrt(n))
            feat = feat.reshape(b, s, n // s, c).permute(0, 3, 1, 2)
        if self.resize_ratio > 1:                             # feature_resize pooling (:51-53): avg_pool_kernel on device tensors
            from .postproc import avg_pool
            feat = avg_pool(feat, self.resize_ratio)
        # the reference's TF.normalize(mean=0,std=1) clone (:56) and fp16 cast (:59-60) are already
        # materialised by the kernels: `feat` is a freshly written fp16 buffer owned by this call.
        feat = feat.detach()
        if self.accept_all:                                   # (:65-66)
            feat = feat.cpu()
        if self.store_idx is None:
            self.feats[feat_id] = feat
        else:                                                 # background extraction (:70-76)
            entry = self.feats[feat_id] if feat_id in self.feats else {'feat': {}, 'count': 0}
            current_idx = entry['count'] + 1
            if current_idx in self.store_idx:
                entry['feat'][current_idx] = feat
            entry['count'] = current_idx
            self.feats[feat_id] = entry

    @property
    def stored_feats(self):
        return self.feats


class FeatureGatherer:
    """Kept for API compatibility (reference :83-89): forwards `module_id-feat_id` to the store."""

    def __init__(self, module_id, feature_store):
        self.module_id = module_id
        self.feature_store = feature_store

    def gather(self, feat, feat_id):
        self.feature_store.store(feat, '-'.join([self.module_id, feat_id]))


def unet_layer_ids(cfg, include_dropped=False):
    """Every hook id of a UNet architecture in execution order — the id scheme of the reference's
    prepare_feature_extractor (:126-249) combined with the gather sites inside the model files.
    With include_dropped=False this equals the key order of feature/configs/config_*_full.json."""
    boc = cfg["block_out_channels"]
    L, nl = len(boc), cfg["layers_per_block"]
    ids = ["unet-in", "unet-after-conv-in"]

    def res(m):
        ids.extend([f"{m}-res-increment", f"{m}-res-out"])

    def vit(m, depth):
        for i in range(depth):
            b = f"{m}-block{i}"
            ids.extend([f"{b}-self-q", f"{b}-self-k", f"{b}-self-v", f"{b}-self-map", f"{b}-cross-q"])
            if include_dropped:
                ids.extend([f"{b}-cross-k", f"{b}-cross-v"])
            ids.extend([f"{b}-cross-map", f"{b}-ffn-inner", f"{b}-out"])
        ids.append(f"{m}-out")

    for lv in range(L):
        for r in range(nl):
            res(f"down-level{lv}-repeat{r}")
            if cfg["has_attn"][lv]:
                vit(f"down-level{lv}-repeat{r}-vit", cfg["transformer_layers"][lv])
        if lv != L - 1:
            ids.append(f"down-level{lv}-downsampler-out")
    res("mid-repeat0")
    vit("mid-vit", cfg["transformer_layers"][-1])
    res("mid-repeat1")
    for i in range(L):
        lv = L - 1 - i
        for r in range(nl + 1):
            res(f"up-level{i}-repeat{r}")
            if cfg["has_attn"][lv]:
                vit(f"up-level{i}-repeat{r}-vit", cfg["transformer_layers"][lv])
        if i != L - 1:
            ids.append(f"up-level{i}-upsampler-out")
    ids.append("unet-out")
    return ids


def flux_layer_ids(cfg):
    """Every hook id of a Flux MMDiT in execution order: the flux branch of the reference's prepare_feature_extractor
    (:98-123) combined with the gather sites (transformer_flux.py:107-108,196-207; attention_processor.py:2280-2291,
    2355-2361; attention.py:1255-1257; the `*-map` ids come from the eager FluxAttnStoreProcessor, components/attention.py:
    493-502, which the reference installs as soon as one map id is requested).  Single blocks continue the numbering."""
    ids = []
    for i in range(cfg["num_layers"]):
        b = f"vit-block{i}"
        ids += [f"{b}-q", f"{b}-k", f"{b}-v", f"{b}-cross-map", f"{b}-self-map", f"{b}-attn-out", f"{b}-norm-out",
                f"{b}-ffn-inner", f"{b}-out"]
    for j in range(cfg["num_single_layers"]):
        b = f"vit-block{cfg['num_layers'] + j}"
        ids += [f"{b}-q", f"{b}-k", f"{b}-v", f"{b}-cross-map", f"{b}-self-map", f"{b}-attn-out", f"{b}-out"]
    return ids


def dit_layer_ids(cfg, include_dropped=False):
    """Hook ids of a PixArt DiT in execution order (reference prepare_feature_extractor :250-286 + gather sites in
    attention.py:589-590,1255-1257 and attention_processor.py:3291-3294; cross-k / cross-v are dropped by the store)."""
    ids = []
    for i in range(cfg["num_layers"]):
        b = f"vit-block{i}"
        ids += [f"{b}-self-q", f"{b}-self-k", f"{b}-self-v", f"{b}-self-map", f"{b}-cross-q"]
        if include_dropped:
            ids += [f"{b}-cross-k", f"{b}-cross-v"]
        ids += [f"{b}-cross-map", f"{b}-ffn-inner", f"{b}-out"]
    return ids


def prepare_feature_extractor(version, pipe, config, resize_ratio, train_unet):
    """Same signature as the reference (:92).  `config`: JSON path, dict, or None/{} (= accept all)."""
    if isinstance(config, str):
        with open(config, 'r') as f:
            config = json.load(f)
    feature_store = FeatureStore(config, resize_ratio, train_unet)
    if version == 'flux':                          # reference :98-123
        if not hasattr(pipe.transformer, 'forward_raw'):
            raise NotImplementedError("pipe.transformer is not the native MMDiT (components.native.NativeFluxTransformer)")
        pipe.transformer.feature_store = feature_store
        return feature_store
    if hasattr(pipe, 'transformer'):               # DiT branch of the reference (:250-286): PixArt alpha / sigma
        if not hasattr(pipe.transformer, 'forward_raw'):
            raise NotImplementedError("pipe.transformer is not a native DiT (components.native.NativePixArtTransformer)")
        pipe.transformer.feature_store = feature_store
        return feature_store
    pipe.unet.feature_store = feature_store       # the native UNet delivers hook tensors here
    return feature_store


ATTENTION_CATEGORIES = ('down_cross', 'mid_cross', 'up_cross', 'down_self', 'mid_self', 'up_self')


def layer_grid(cfg, layer_id, lat):
    """Spatial size (tokens per side) of the UNet level a layer id belongs to, for a lat x lat latent."""
    L = len(cfg["block_out_channels"])
    parts = layer_id.split('-')
    if parts[0] == 'mid':
        lv = L - 1
    else:
        lv = int(parts[1][5:])
        if parts[0] == 'up':
            lv = L - 1 - lv
    return lat >> lv


def attention_map_ids(cfg, all_ids, categories, lat, min_size, max_size):
    """'*-map' ids feeding the aggregated `attention=[...]` feature: the reference's AttentionStore keeps the maps whose
    query grid lies in [min_size, max_size] (components/attention.py:109-115, sizes img/32 .. img/16 at :541)."""
    out = {c: [] for c in categories}
    for i in all_ids:
        if not i.endswith('-map'):
            continue
        place = i.split('-')[0]
        kind = 'cross' if i.endswith('-cross-map') else 'self'
        key = f"{place}_{kind}"
        if key in out and min_size <= layer_grid(cfg, i, lat) <= max_size:
            out[key].append(i)
    return out


def dit_attention_map_ids(all_ids, categories, grid, min_size, max_size):
    """PixArt / DiT: the reference registers every attention layer with place_in_unet='up' and an AttentionStore(img/32, img/8)
    (components/attention.py:567-590), so `up_cross` / `up_self` collect every block's map when the token grid is in range."""
    out = {c: [] for c in categories}
    if min_size <= grid <= max_size:
        for i in all_ids:
            if i.endswith('-cross-map') and 'up_cross' in out:
                out['up_cross'].append(i)
            elif i.endswith('-self-map') and 'up_self' in out:
                out['up_self'].append(i)
    return out


def aggregate_attention(maps_by_category, out_size):
    """Reference AttentionStore.aggregate_attention (components/attention.py:141-161) + diffusion_feature.py:492-500:
    head-mean maps (B,Q,K) -> (B,K,h,w), averaged over the layers of the same category and size, nearest-resized to
    out_size and concatenated over the channel dim, in category order then first-seen size order.  Device tensors run on
    maps_mean_kernel + resize_concat_kernel (components/postproc.py)."""
    from .postproc import aggregate_maps
    return aggregate_maps(maps_by_category, out_size)
