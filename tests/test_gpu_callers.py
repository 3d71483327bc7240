"""The callers on the far side of the path (-m gpu; SURVEY §8 "next": the evaluators that consume `FeatureExtractor`), restated as the few lines
with which they drive it — the extractor must drop in "unchanged" (BASELINE.json north_star):

  * correspondence/correspondence/aggregation_network.py:34-66   several extractors (own layer set / version / attention categories / image size),
    `encode_prompt` once, `offload_prompt_encoder(persistent=True)`, `extract(prompts=, batch_size=1, image=[PIL], t=)`, then
    `F.interpolate(f, (128, 128), mode='bilinear').squeeze()` over `feat.values()`, `torch.cat`, `.type(torch.float32)`, a conv head;
  * segmentation/models/diffusion_segmentor.py:215-262            `extract(prompts=, batch_size=B, image=<tensor>, image_type='tensors', t=, use_control=False)`,
    `features[layer].type(torch.float32)` into per-layer convs, a training step on those convs.
What is asserted is interoperability — the returned tensors are channels-last VIEWS of plan-owned buffers: torch's interpolate / cat / conv / autograd
must treat them exactly like the contiguous NCHW tensors the reference returns, and they must stay valid while other extractors run.
(The values themselves are pinned against the oracle in tests/test_gpu_fullsize.py / test_gpu_unet.py.)
"""
import numpy as np
import pytest
import torch
import torch.nn as nn
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _pil(size, seed):
    from PIL import Image
    return Image.fromarray((np.random.RandomState(seed).rand(size, size, 3) * 255).astype(np.uint8))


def test_correspondence_aggregation_network_call_pattern(monkeypatch):
    monkeypatch.setenv("GDF_SYNTHETIC_WEIGHTS", "1")
    from diffusion_feature import FeatureExtractor
    device = "cuda:0"
    configs = [dict(layer={"up-level1-repeat1-vit-block0-cross-q": True, "up-level2-repeat1-res-out": True}, version="1-5", attention=["up_cross"], img_size=256, t=100),
               dict(layer={"up-level1-repeat0-vit-block0-out": True, "mid-vit-block0-ffn-inner": True}, version="2-1", attention=None, img_size=256, t=261)]
    prompt = "a photo of a cat"
    extractors = []
    for config in configs:                                                   # aggregation_network.py:34-48
        fe = FeatureExtractor(layer=config["layer"], version=config["version"], device=device, attention=config["attention"], img_size=config["img_size"])
        extractors.append({"model": fe, "prompt_embeds": fe.encode_prompt(prompt), "t": config["t"]})
        fe.offload_prompt_encoder(persistent=True)
    img = _pil(300, 0)                                                       # any size: extract() resizes to img_size
    held, features = [], []
    for ex in extractors:                                                    # :52-63
        feat = ex["model"].extract(prompts=ex["prompt_embeds"], batch_size=1, image=[img], t=ex["t"])
        held.append((feat, {k: v.clone() for k, v in feat.items()}))
        for f in feat.values():
            features.append(F.interpolate(f, (128, 128), mode="bilinear").squeeze())
    x = torch.cat(features, dim=0)
    # same numbers as through contiguous copies of the features (the reference hands out contiguous NCHW tensors)
    want = torch.cat([F.interpolate(c.contiguous(), (128, 128), mode="bilinear").squeeze() for _, cl in held for c in cl.values()], dim=0)
    assert x.dtype == torch.float16 and x.shape == want.shape and x.shape[1:] == (128, 128) and torch.equal(x, want)
    assert "attn" in held[0][0] and held[0][0]["attn"].shape[0] == 1 and "attn" not in held[1][0]        # the aggregated attention feature rides in the dict (:492-500)
    assert list(held[0][0].keys())[:2] == ["up-level1-repeat1-vit-block0-cross-q", "up-level2-repeat1-res-out"]
    # the first extractor's features are still intact after the second one ran (and after a further call of the second)
    extractors[1]["model"].extract(prompts=extractors[1]["prompt_embeds"], batch_size=1, image=[_pil(300, 1)], t=261)
    torch.cuda.synchronize()
    for feat, cl in held:
        for k in cl:
            assert torch.equal(feat[k], cl[k]), k
    head = nn.Conv2d(x.shape[0], 8, 1).to(device)                            # :64-66  x.type(torch.float32) -> self.out(x)
    y = head(x.type(torch.float32)[None])
    assert y.shape == (1, 8, 128, 128) and torch.isfinite(y).all()


def test_segmentor_call_pattern_with_a_training_step_on_the_heads(monkeypatch):
    monkeypatch.setenv("GDF_SYNTHETIC_WEIGHTS", "1")
    from diffusion_feature import FeatureExtractor
    device = "cuda:0"
    layers = [["up-level1-repeat1-vit-block0-out", "up-level1-repeat2-res-out"], ["up-level2-repeat0-res-out"]]      # two resolution levels
    fe = FeatureExtractor(layer={k: True for lv in layers for k in lv}, version="1-5", device=device, img_size=256)
    prompt_embeds = fe.encode_prompt("a street scene")
    fe.offload_prompt_encoder(persistent=True)
    inputs = torch.rand(3, 3, 200, 200, device=device) * 2 - 1               # diffusion_segmentor.py: normalised image tensors, any size
    convs = nn.ModuleDict()
    opt = None
    for step in range(2):
        features = fe.extract(prompts=prompt_embeds, batch_size=inputs.shape[0], image=inputs, image_type="tensors", t=100, use_control=False)
        if not convs:
            for lv in layers:
                for k in lv:
                    convs[k.replace("-", "_")] = nn.Conv2d(features[k].shape[1], 16, 1).to(device)
            opt = torch.optim.SGD(convs.parameters(), lr=1e-3)
        outs = []
        for lv in layers:                                                    # :236-250
            per_level = [convs[k.replace("-", "_")](features[k].type(torch.float32)) for k in lv]
            outs.append(torch.cat(per_level, dim=1))
        loss = sum(o.square().mean() for o in outs)
        opt.zero_grad(); loss.backward(); opt.step()
        assert all(p.grad is not None and torch.isfinite(p.grad).all() for p in convs.parameters())
        for k in features:
            assert features[k].shape[0] == 3 and features[k].dtype == torch.float16 and not features[k].requires_grad
    assert outs[0].shape[1] == 32 and outs[1].shape[1] == 16 and outs[0].shape[-1] * 2 == outs[1].shape[-1]
