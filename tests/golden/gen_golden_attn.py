"""Golden vector of the aggregated attention feature from the REFERENCE's own AttentionStore (build container only).

    python tests/golden/gen_golden_attn.py       # needs /root/reference; writes tests/golden/attn_aggregate.npz

Random probability maps (B, heads, Q, K) are pushed through the reference's AttentionStore exactly as its
AttnStoreProcessor does (components/attention.py:238-244: `store(attention_probs.mean(1), is_cross, place)`), then
aggregate_attention (:141-161) and the interpolate / cat of diffusion_feature.py:492-500.  Pure data in, pure data out."""
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle import ref_blocks as RB  # noqa: E402


def main():
    RB.attn_store_processor()
    AttentionStore = sys.modules["gdf_ref_attention"].AttentionStore
    g = torch.Generator().manual_seed(0)
    # (hook id, Q, K): execution order of a 3-level toy UNet; 2x2 (too small) and 10x10 (too large) grids must be dropped
    layout = [("down-level0-repeat0-vit-block0-self-map", 100, 100), ("down-level0-repeat0-vit-block0-cross-map", 100, 7),
              ("down-level1-repeat0-vit-block0-self-map", 64, 64), ("down-level1-repeat0-vit-block0-cross-map", 64, 7),
              ("down-level1-repeat1-vit-block0-cross-map", 64, 7), ("mid-vit-block0-cross-map", 4, 7),
              ("up-level0-repeat0-vit-block0-cross-map", 16, 7), ("up-level0-repeat1-vit-block0-self-map", 16, 16),
              ("up-level0-repeat1-vit-block0-cross-map", 16, 7), ("up-level1-repeat0-vit-block0-cross-map", 64, 7),
              ("up-level1-repeat1-vit-block0-cross-map", 64, 7), ("up-level1-repeat2-vit-block1-cross-map", 64, 7),
              ("up-level1-repeat2-vit-block1-self-map", 64, 64)]
    store = AttentionStore(min_size=4, max_size=8)
    arrs = {}
    for hid, q, k in layout:
        probs = torch.softmax(2.0 * torch.randn(2, 3, q, k, generator=g), -1)
        arrs["map:" + hid] = probs.half().numpy()
        probs = probs.half().float()
        place = hid.split("-")[0]
        store(probs.mean(1), hid.endswith("-cross-map"), place)
    for name, sel in (("a", ["up_cross", "down_self"]), ("b", ["down_cross", "up_self", "up_cross"])):
        ref = store.aggregate_attention(sel)
        out = torch.cat([F.interpolate(a, size=(16, 16)) for c in sel for a in ref[c].values()], dim=-3)
        arrs["out:" + name] = out.numpy()
        arrs["sel:" + name] = np.array(repr(sel))
    arrs["meta"] = np.array(repr(dict(order=[h for h, _, _ in layout], min_size=4, max_size=8, out_size=16)))
    path = os.path.join(HERE, "attn_aggregate.npz")
    np.savez_compressed(path, **arrs)
    print({k: v.shape for k, v in arrs.items() if k.startswith("out:")}, os.path.getsize(path) // 1024, "KiB")


if __name__ == "__main__":
    main()
