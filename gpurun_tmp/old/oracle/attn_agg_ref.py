"""CPU ORACLE (test infrastructure only): the aggregated attention feature `feats['attn']`.

Restates, on '-map' hook tensors (B, heads, Q, K):
  * AttnStoreProcessor handing `attention_probs.mean(1)` to the store          feature/components/attention.py:238-244
  * AttentionStore.forward: keep maps with min_size^2 <= Q <= max_size^2,        feature/components/attention.py:109-115
    keyed `{down|mid|up}_{cross|self}`, in execution order
  * AttentionStore.aggregate_attention: group by sqrt(Q), mean over the group    feature/components/attention.py:141-161
  * FeatureExtractor.extract: nearest-resize every group to img/8 and concat     feature/diffusion_feature.py:492-500
    over channels (category order of the `attention=[...]` list, then first-seen size order); min / max size = img/32,
    img/16 (:541 of components/attention.py via diffusion_feature.py:67-68)
Pinned by tests/golden/attn_aggregate.npz (gen_golden_attn.py: the reference's own AttentionStore on random maps)."""
import math

import torch
import torch.nn.functional as F


def category_of(map_id):
    place = map_id.split("-")[0]
    return f"{place}_{'cross' if map_id.endswith('-cross-map') else 'self'}"


def aggregate(maps_in_order, selector, min_size, max_size, out_size):
    """maps_in_order: [(hook id, (B, heads, Q, K) tensor)] in execution order -> (B, sum K, out_size, out_size) fp32"""
    store = {}
    for hid, m in maps_in_order:
        a = m.float().mean(1)                                   # (B, Q, K)
        if min_size ** 2 <= a.shape[1] <= max_size ** 2:
            store.setdefault(category_of(hid), []).append(a)
    outs = []
    for cat in selector:
        by_size = {}
        for a in store.get(cat, []):
            size = int(math.sqrt(a.shape[1]))
            b, q, k = a.shape
            by_size.setdefault(size, []).append(a.reshape(b, size, q // size, k).permute(0, 3, 1, 2))
        for size, lst in by_size.items():
            outs.append(F.interpolate(torch.stack(lst).mean(0), size=(out_size, out_size)))
    return torch.cat(outs, dim=-3)
