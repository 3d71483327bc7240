"""Generate the PixArt DiT golden vector from the REFERENCE's own modules (build container only).

    python tests/golden/gen_golden_pixart.py     # needs /root/reference; writes tests/golden/pixart_tiny.npz

Runs the reference's Transformer2DModel (feature/diffusers/models/transformers/transformer_2d.py, patched-input +
ada_norm_single branch) with its BasicTransformerBlock / FeedForward (attention.py), Attention + AttnProcessor2_0
(attention_processor.py) and FeatureStore / FeatureGatherer, wired with the DiT ids of
components/feature_extractor.py:268-286, on a tiny DiT (2 blocks, 8 heads x 72) with a ragged text mask.  Un-vendored
classes (PatchEmbed, AdaLayerNormSingle, PixArtAlphaTextProjection, GELU) come from the scaffolding in
oracle/ref_blocks.py.  Fixture = inputs, per-tensor weight checksums (weights = synth_params seed 5), output, every hook."""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle import pixart_ref as PR  # noqa: E402
from oracle import ref_blocks as RB  # noqa: E402


@torch.no_grad()
def main():
    m = RB.modules()
    arch = PR.tiny_arch()
    P = PR.synth_params(arch, seed=5)
    I = PR.synth_inputs(arch, batch=2, lat=8, n_txt=12, seed=6, valid=[12, 7])
    model = m.Transformer2DModel(
        num_attention_heads=arch["num_attention_heads"], attention_head_dim=arch["attention_head_dim"],
        in_channels=arch["in_channels"], out_channels=arch["out_channels"], num_layers=arch["num_layers"],
        cross_attention_dim=PR.inner_dim(arch), attention_bias=True, sample_size=arch["sample_size"],
        patch_size=arch["patch_size"], activation_fn="gelu-approximate", norm_type="ada_norm_single",
        norm_elementwise_affine=False, norm_eps=1e-6, caption_channels=arch["caption_channels"],
        interpolation_scale=arch["interpolation_scale"], use_additional_conditions=False)
    missing = model.load_state_dict(P, strict=False)
    assert not missing.unexpected_keys and all("pos_embed.pos_embed" in k for k in missing.missing_keys), missing
    store = m.FeatureStore({}, 1, True)
    G = m.FeatureGatherer
    for i, bb in enumerate(model.transformer_blocks):          # components/feature_extractor.py:268-286
        bb.feature_gatherer = G(f"vit-block{i}", store)
        bb.attn1.feature_gatherer = G(f"vit-block{i}-self", store)
        bb.attn2.feature_gatherer = G(f"vit-block{i}-cross", store)
        bb.ff.feature_gatherer = G(f"vit-block{i}-ffn", store)
    y = model(I["hidden_states"], encoder_hidden_states=I["encoder_hidden_states"], timestep=I["timestep"],
              encoder_attention_mask=I["encoder_attention_mask"], added_cond_kwargs={"resolution": None, "aspect_ratio": None},
              return_dict=False)[0]
    arrs = {"wsum": np.array([float(v.double().sum()) for v in P.values()]),
            "wabs": np.array([float(v.double().abs().sum()) for v in P.values()])}
    for k, v in I.items():
        arrs["in:" + k] = v.numpy().astype(np.float32)
    arrs["out:y"] = y.float().numpy()
    for k, v in store.stored_feats.items():
        arrs["out:hook:" + k] = v.detach().float().numpy()
    arrs["meta"] = np.array(repr(dict(arch=arch, wseed=5, order=list(store.stored_feats.keys()))))
    path = os.path.join(HERE, "pixart_tiny.npz")
    np.savez_compressed(path, **arrs)
    print("pixart_tiny ->", os.path.getsize(path) // 1024, "KiB;", len(store.stored_feats), "hooks")
    st = PR.Store(None, out_dtype=None)
    y2 = PR.pixart_forward(P, arch, I["hidden_states"], I["encoder_hidden_states"], I["timestep"], I["encoder_attention_mask"], st,
                           want_map=False)
    assert list(st.feats.keys()) == list(store.stored_feats.keys()), (list(st.feats.keys()), list(store.stored_feats.keys()))
    worst = float((y2 - y).abs().max())
    for k in st.feats:
        worst = max(worst, float((st.feats[k].float() - store.stored_feats[k].float()).abs().max()))
    print("oracle vs reference: max abs diff", worst)
    assert worst < 2e-4, worst

    # ---- second fixture: every attention on the reference's eager AttnStoreProcessor (components/attention.py:176-263,
    # installed for DiTs by register_attention_store, :582-590) -> `self-map` / `cross-map` hooks with the ragged mask
    Proc = RB.attn_store_processor()
    store = m.FeatureStore({}, 1, True)
    for i, bb in enumerate(model.transformer_blocks):
        bb.feature_gatherer = G(f"vit-block{i}", store)
        bb.attn1.feature_gatherer = G(f"vit-block{i}-self", store)
        bb.attn2.feature_gatherer = G(f"vit-block{i}-cross", store)
        bb.ff.feature_gatherer = G(f"vit-block{i}-ffn", store)
        bb.attn1.processor = Proc(attnstore=None, place_in_unet="up")
        bb.attn2.processor = Proc(attnstore=None, place_in_unet="up")
    ym = model(I["hidden_states"], encoder_hidden_states=I["encoder_hidden_states"], timestep=I["timestep"],
               encoder_attention_mask=I["encoder_attention_mask"], added_cond_kwargs={"resolution": None, "aspect_ratio": None},
               return_dict=False)[0]
    arrs = {"out:y": ym.float().numpy()}
    for k, v in store.stored_feats.items():
        if k.endswith("-map"):
            arrs["out:hook:" + k] = v.detach().float().numpy()
    arrs["meta"] = np.array(repr(dict(arch=arch, wseed=5, order=list(store.stored_feats.keys()))))
    path = os.path.join(HERE, "pixart_tiny_maps.npz")
    np.savez_compressed(path, **arrs)
    print("pixart_tiny_maps ->", os.path.getsize(path) // 1024, "KiB")
    st = PR.Store(None, out_dtype=None)
    y3 = PR.pixart_forward(P, arch, I["hidden_states"], I["encoder_hidden_states"], I["timestep"], I["encoder_attention_mask"], st)
    assert list(st.feats.keys()) == list(store.stored_feats.keys()) == PR.hook_ids(arch, maps=True), list(store.stored_feats.keys())
    worst = float((y3 - ym).abs().max())
    for k in st.feats:
        worst = max(worst, float((st.feats[k].float() - store.stored_feats[k].float()).abs().max()))
    print("oracle (maps) vs reference: max abs diff", worst)
    assert worst < 2e-4, worst


if __name__ == "__main__":
    if not RB.available():
        sys.exit("reference tree not found; goldens can only be generated in the build container")
    main()
