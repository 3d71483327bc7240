"""Generate the Flux MMDiT golden vector from the REFERENCE's own modules (build container only).

    python tests/golden/gen_golden_flux.py     # needs /root/reference; writes tests/golden/flux_tiny.npz

Runs the reference's FluxTransformer2DModel.forward (feature/diffusers/models/transformers/transformer_flux.py:414-603)
with its FluxTransformerBlock / FluxSingleTransformerBlock, FluxAttnProcessor2_0 (attention_processor.py:2259-2362),
FeedForward (attention.py:1195-1258) and FeatureStore/FeatureGatherer (components/feature_extractor.py), wired with the
same gatherer ids as prepare_feature_extractor's flux branch (feature_extractor.py:98-123), on a tiny MMDiT
(2 double + 2 single blocks, 2 heads x 128).  The un-vendored diffusers classes those files import
(AdaLayerNormZero*, RMSNorm, FluxPosEmbed, apply_rotary_emb, Combined*Embeddings, GELU) come from the scaffolding in
oracle/ref_blocks.py.  The fixture is pure data: inputs, per-tensor weight checksums (weights = synth_params seed 3), fp32 output + every hook.
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle import flux_ref as FR  # noqa: E402
from oracle import ref_blocks as RB  # noqa: E402


@torch.no_grad()
def main():
    m = RB.modules()
    arch = FR.tiny_arch()
    P = FR.synth_params(arch, seed=3)
    I = FR.synth_inputs(arch, batch=2, grid=4, n_txt=8, seed=4, same_prompt=False)
    model = m.FluxTransformer2DModel(
        patch_size=1, in_channels=arch["in_channels"], num_layers=arch["num_layers"],
        num_single_layers=arch["num_single_layers"], attention_head_dim=arch["attention_head_dim"],
        num_attention_heads=arch["num_attention_heads"], joint_attention_dim=arch["joint_attention_dim"],
        pooled_projection_dim=arch["pooled_projection_dim"], guidance_embeds=arch["guidance_embeds"],
        axes_dims_rope=arch["axes_dims_rope"])
    missing, unexpected = model.load_state_dict(P, strict=True), None
    store = m.FeatureStore({}, 1, True)                       # accept-all; train_unet=True -> no .to('cuda')
    G = m.FeatureGatherer
    # same assignment as components/feature_extractor.py:98-123
    model.feature_gatherer = G("vit", store)
    i = -1
    for i, blk in enumerate(model.transformer_blocks):
        blk.feature_gatherer = G(f"vit-block{i}", store)
        blk.attn.feature_gatherer = G(f"vit-block{i}", store)
        blk.ff.feature_gatherer = G(f"vit-block{i}-ffn", store)
    for blk in model.single_transformer_blocks:
        i += 1
        blk.feature_gatherer = G(f"vit-block{i}", store)
        blk.attn.feature_gatherer = G(f"vit-block{i}", store)
    y = model(hidden_states=I["hidden_states"], encoder_hidden_states=I["encoder_hidden_states"],
              pooled_projections=I["pooled_projections"], timestep=I["timestep"], img_ids=I["img_ids"],
              txt_ids=I["txt_ids"], guidance=I["guidance"], return_dict=False)[0]
    # weights are NOT stored (36 C^2 parameters per block do not compress): the fixture pins them by seed + checksums
    # of oracle.flux_ref.synth_params(arch, seed=3) (pure CPU torch RNG)
    arrs = {"wsum": np.array([float(v.double().sum()) for v in P.values()]),
            "wabs": np.array([float(v.double().abs().sum()) for v in P.values()])}
    for k, v in I.items():
        if v is not None:
            arrs["in:" + k] = v.numpy().astype(np.float32)
    arrs["out:y"] = y.float().numpy()
    for k, v in store.stored_feats.items():
        arrs["out:hook:" + k] = v.detach().float().numpy()
    arrs["meta"] = np.array(repr(dict(arch=arch, wseed=3, order=list(store.stored_feats.keys()))))
    path = os.path.join(HERE, "flux_tiny.npz")
    np.savez_compressed(path, **arrs)
    print("flux_tiny ->", os.path.getsize(path) // 1024, "KiB;", len(store.stored_feats), "hooks")

    # self-check of the restatement against the reference run
    st = FR.Store(None, out_dtype=None)
    y2 = FR.flux_forward(P, arch, I["hidden_states"], I["encoder_hidden_states"], I["pooled_projections"], I["timestep"],
                         I["img_ids"], I["txt_ids"], I["guidance"], store=st, want_map=False)
    assert list(st.feats.keys()) == list(store.stored_feats.keys()), "hook order differs"
    worst = float((y2 - y).abs().max())
    for k in st.feats:
        worst = max(worst, float((st.feats[k].float() - store.stored_feats[k].float()).abs().max()))
    print("oracle vs reference: max abs diff", worst)
    assert worst < 2e-4, worst


@torch.no_grad()
def main_maps():
    """Second fixture: every attention of the model on the reference's FluxAttnStoreProcessor (components/attention.py:404-527,
    installed by register_attention_store(..., processor_only=True), :562-596) -> `cross-map` / `self-map` hooks."""
    m = RB.modules()
    Proc = RB.flux_attn_store_processor()
    arch = FR.tiny_arch()
    P = FR.synth_params(arch, seed=3)
    I = FR.synth_inputs(arch, batch=2, grid=4, n_txt=8, seed=4, same_prompt=False)
    model = m.FluxTransformer2DModel(
        patch_size=1, in_channels=arch["in_channels"], num_layers=arch["num_layers"],
        num_single_layers=arch["num_single_layers"], attention_head_dim=arch["attention_head_dim"],
        num_attention_heads=arch["num_attention_heads"], joint_attention_dim=arch["joint_attention_dim"],
        pooled_projection_dim=arch["pooled_projection_dim"], guidance_embeds=arch["guidance_embeds"],
        axes_dims_rope=arch["axes_dims_rope"])
    model.load_state_dict(P, strict=True)
    store = m.FeatureStore({}, 1, True)
    G = m.FeatureGatherer
    model.feature_gatherer = G("vit", store)
    i = -1
    for i, blk in enumerate(model.transformer_blocks):
        blk.feature_gatherer = G(f"vit-block{i}", store); blk.attn.feature_gatherer = G(f"vit-block{i}", store)
        blk.ff.feature_gatherer = G(f"vit-block{i}-ffn", store)
        blk.attn.processor = Proc(attnstore=None, place_in_unet="up")
    for blk in model.single_transformer_blocks:
        i += 1
        blk.feature_gatherer = G(f"vit-block{i}", store); blk.attn.feature_gatherer = G(f"vit-block{i}", store)
        blk.attn.processor = Proc(attnstore=None, place_in_unet="up")
    y = model(hidden_states=I["hidden_states"], encoder_hidden_states=I["encoder_hidden_states"],
              pooled_projections=I["pooled_projections"], timestep=I["timestep"], img_ids=I["img_ids"],
              txt_ids=I["txt_ids"], guidance=I["guidance"], return_dict=False)[0]
    arrs = {"out:y": y.float().numpy()}
    for k, v in store.stored_feats.items():
        if k.endswith("-map"):
            arrs["out:hook:" + k] = v.detach().float().numpy()
    arrs["meta"] = np.array(repr(dict(arch=arch, wseed=3, order=list(store.stored_feats.keys()))))
    path = os.path.join(HERE, "flux_tiny_maps.npz")
    np.savez_compressed(path, **arrs)
    print("flux_tiny_maps ->", os.path.getsize(path) // 1024, "KiB;", sum(k.endswith("-map") for k in store.stored_feats), "maps")
    st = FR.Store(None, out_dtype=None)
    y2 = FR.flux_forward(P, arch, I["hidden_states"], I["encoder_hidden_states"], I["pooled_projections"], I["timestep"],
                         I["img_ids"], I["txt_ids"], I["guidance"], store=st, want_map=True)
    assert list(st.feats.keys()) == list(store.stored_feats.keys()) == FR.hook_ids(arch, maps=True)
    worst = float((y2 - y).abs().max())
    for k in st.feats:
        worst = max(worst, float((st.feats[k].float() - store.stored_feats[k].float()).abs().max()))
    print("oracle (maps) vs reference: max abs diff", worst)
    assert worst < 2e-4, worst


if __name__ == "__main__":
    if not RB.available():
        sys.exit("reference tree not found; goldens can only be generated in the build container")
    main()
    main_maps()
