"""GPU parity of the PixArt DiT path (include/gdf_pixart.h) through the C ABI: whole tiny DiT with every
hook vs the CPU oracle (oracle/pixart_ref.py) and vs the committed reference golden (tests/golden/pixart_tiny.npz, produced
by the reference's own Transformer2DModel).  Stated tolerance: relative L2 error per hooked tensor <= 3e-3."""
import ast
import os

import numpy as np
import pytest
import torch

from helpers import rel_l2
from oracle import pixart_ref as PR

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")
TOL = 3e-3


def _run(arch, P, I, ids):
    from components.native import NativePixArtTransformer
    net = NativePixArtTransformer(arch, device="cuda:0")
    net.load_state_dict({k: v.half() for k, v in P.items()})
    assert net.ready()
    out, hooks = net.forward_raw(I["hidden_states"].cuda(), I["encoder_hidden_states"].cuda(), I["timestep"].cuda(),
                                 I["encoder_attention_mask"].cuda(), hook_ids=ids)
    torch.cuda.synchronize()
    return net, out, hooks


def test_sincos_pos_embed_and_patch_roundtrip():
    import ctypes
    from ops_binding import P, lib, ok, stream
    L = lib()
    for gh, gw, C, base, isc in ((4, 4, 144, 8, 1.0), (64, 64, 1152, 64, 2.0), (6, 10, 288, 8, 1.0)):
        out = torch.empty(gh * gw, C, device="cuda")
        ok(L.gdf_op_sincos_pos_embed(P(out), C, gh, gw, base, ctypes.c_float(isc), stream()), L)
        ref = PR.sincos_pos_embed(C, gh, gw, base, isc)
        assert torch.allclose(out.cpu(), ref, atol=2e-6), (gh, gw, float((out.cpu() - ref).abs().max()))


@pytest.mark.parametrize("heads,lat,n_txt,valid", [(8, 16, 24, [24, 9]), (16, 8, 300, [300, 77])])
def test_pixart_tiny_all_hooks_vs_oracle(heads, lat, n_txt, valid):
    arch = PR.tiny_arch(heads=heads)
    P = PR.synth_params(arch, seed=0)
    I = PR.synth_inputs(arch, 2, lat, n_txt, seed=1, valid=valid)
    st = PR.Store(None)                                   # accept-all: eager processor incl. `*-map` hooks, like the reference
    y = PR.pixart_forward(P, arch, I["hidden_states"], I["encoder_hidden_states"], I["timestep"], I["encoder_attention_mask"], st)
    net, out, hooks = _run(arch, P, I, PR.hook_ids(arch, maps=True))
    assert net.hook_names() == PR.hook_ids(arch, maps=True) and list(hooks.keys()) == list(st.feats.keys())
    assert hooks["vit-block0-cross-map"].shape == (2, heads, (lat // 2) ** 2, n_txt)
    assert out.shape == y.shape and rel_l2(out, y) < TOL, rel_l2(out, y)
    for k, ref in st.feats.items():
        assert hooks[k].shape == ref.shape and hooks[k].dtype == torch.float16, k
        assert rel_l2(hooks[k], ref) < TOL, (k, rel_l2(hooks[k], ref))


def test_pixart_matches_reference_golden():
    z = np.load(os.path.join(GOLD, "pixart_tiny.npz"))
    meta = ast.literal_eval(str(z["meta"]))
    arch = meta["arch"]
    P = PR.synth_params(arch, seed=meta["wseed"])
    I = {k[3:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("in:")}
    net, out, hooks = _run(arch, P, I, meta["order"])
    assert list(hooks.keys()) == meta["order"]
    assert rel_l2(out, torch.from_numpy(z["out:y"])) < TOL
    for k in meta["order"]:
        assert rel_l2(hooks[k], torch.from_numpy(z["out:hook:" + k])) < TOL, k
    zm = np.load(os.path.join(GOLD, "pixart_tiny_maps.npz"))          # the reference on its AttnStoreProcessor (ragged mask)
    mm = ast.literal_eval(str(zm["meta"]))
    net, out, hooks = _run(arch, P, I, mm["order"])
    assert list(hooks.keys()) == mm["order"] == net.hook_names()
    for k in mm["order"]:
        if k.endswith("-map"):
            ref = torch.from_numpy(zm["out:hook:" + k])
            assert hooks[k].shape == ref.shape and rel_l2(hooks[k], ref) < TOL, (k, rel_l2(hooks[k], ref))


@pytest.mark.parametrize("version", ["pixart-sigma", "pixart-alpha"])
def test_feature_extractor_api_pixart_synthetic(version):
    """FeatureExtractor(version='pixart-sigma' | 'pixart-alpha') through the drop-in class (reference diffusion_feature.py:277-283, 466-474):
    native VAE encode + noise-add, native DiT forward, hooks via FeatureStore."""
    import numpy as np
    from PIL import Image
    import diffusion_feature
    from components.feature_extractor import dit_layer_ids
    from components.models import SyntheticPixartPipe
    arch = PR.tiny_arch(heads=8, num_layers=2, sample_size=16)
    pipe = SyntheticPixartPipe(version, "cuda:0", seed=0, cfg=arch, n_txt=20)
    assert pipe.transformer.hook_names() == dit_layer_ids(arch) == PR.hook_ids(arch, maps=True)
    assert pipe.vae.config.scaling_factor == (0.18215 if version == "pixart-alpha" else 0.13025)     # sd-vae-ft-ema vs the SDXL VAE
    layer = {"vit-block1-out": True, "vit-block0-cross-q": True, "vit-block1-ffn-inner": True, "vit-block0-cross-k": True}
    df = diffusion_feature.FeatureExtractor(layer=layer, version=version, img_size=128, device='cuda:0', external_model=pipe)
    prompt = df.encode_prompt('a photo of a cat')
    img = Image.fromarray((np.random.RandomState(0).rand(90, 70, 3) * 255).astype(np.uint8))
    feats = df.extract(prompt, batch_size=2, image=[img, img], t=100)
    assert list(feats.keys()) == ["vit-block0-cross-q", "vit-block1-ffn-inner", "vit-block1-out"]     # cross-k dropped
    assert feats["vit-block1-out"].shape == (2, 576, 8, 8) and feats["vit-block1-ffn-inner"].shape == (2, 2304, 8, 8)
    for v in feats.values():
        assert v.dtype == torch.float16 and v.is_cuda and torch.isfinite(v.float()).all()


def test_pixart_early_exit_and_no_mask():
    from components.native import NativePixArtTransformer
    arch = PR.tiny_arch(heads=8, num_layers=3)
    P = PR.synth_params(arch, seed=2)
    I = PR.synth_inputs(arch, 2, 8, 16, seed=3)                      # all caption tokens valid
    st = PR.Store({"vit-block1-cross-q": True, "vit-block0-out": True})
    PR.pixart_forward(P, arch, I["hidden_states"], I["encoder_hidden_states"], I["timestep"], None, st, want_map=False)
    outs = {}
    for ee in (False, True):
        net = NativePixArtTransformer(arch, device="cuda:0", early_exit=ee)
        net.load_state_dict({k: v.half() for k, v in P.items()})
        _, hooks = net.forward_raw(I["hidden_states"].cuda(), I["encoder_hidden_states"].cuda(), I["timestep"].cuda(), None,
                                   hook_ids=list(st.feats.keys()))
        torch.cuda.synchronize()
        outs[ee] = hooks
    assert list(outs[True].keys()) == list(st.feats.keys())
    for k, ref in st.feats.items():
        assert torch.equal(outs[True][k], outs[False][k]) and rel_l2(outs[True][k], ref) < TOL, k


def test_pixart_aggregated_attention_feature():
    """`attention=[...]` on the DiT branch (reference components/attention.py:567-590: every block registered as 'up',
    AttentionStore(img/32, img/8); diffusion_feature.py:492-500): feats['attn'] = head mean -> mean over the 2 blocks ->
    nearest resize to img/8, checked against the same aggregation of the explicitly hooked maps done with torch."""
    import numpy as np
    import torch.nn.functional as F
    from PIL import Image
    import diffusion_feature
    from components.models import SyntheticPixartPipe
    arch = PR.tiny_arch(heads=8, num_layers=2, sample_size=16)
    pipe = SyntheticPixartPipe("pixart-sigma", "cuda:0", seed=0, cfg=arch, n_txt=24)
    layer = {"vit-block0-cross-map": True, "vit-block1-cross-map": True, "vit-block1-out": True}
    df = diffusion_feature.FeatureExtractor(layer=layer, version='pixart-sigma', img_size=128, device='cuda:0', external_model=pipe,
                                            attention=['up_cross'])
    prompt = df.encode_prompt('a photo of a cat on a mat')
    img = Image.fromarray((np.random.RandomState(0).rand(90, 70, 3) * 255).astype(np.uint8))
    feats = df.extract(prompt, batch_size=2, image=[img, img], t=100)
    assert list(feats.keys()) == ["vit-block0-cross-map", "vit-block1-cross-map", "vit-block1-out", "attn"]
    maps = [feats["vit-block0-cross-map"].float(), feats["vit-block1-cross-map"].float()]        # (B, heads, 64, 24)
    avg = torch.stack([m.mean(1).half().float() for m in maps]).mean(0)                           # (B, 64, 24)
    want = F.interpolate(avg.reshape(2, 8, 8, 24).permute(0, 3, 1, 2), size=(16, 16))
    assert feats["attn"].shape == (2, 24, 16, 16) and feats["attn"].dtype == torch.float16
    assert torch.allclose(feats["attn"].float(), want, atol=1e-3)
