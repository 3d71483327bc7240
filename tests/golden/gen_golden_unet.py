"""Whole-UNet golden vectors from the REFERENCE's own UNet2DConditionModel.forward (build container only).

    python tests/golden/gen_golden_unet.py        # needs /root/reference; writes tests/golden/unet_tiny_{xl,15,21}.npz

oracle/ref_unet.py imports the reference's unet_2d_condition.py (time / text_time embedding path, skip stack, mid block,
conv_norm_out, `unet-*` gather sites) and the reference's prepare_feature_extractor (hook-id scheme) and runs them over the
reference's own ResnetBlock2D / Transformer2DModel / samplers on a shrunken architecture (same topology).  A fixture holds
pure data: the seeded inputs, the ordered hook ids the reference stored, and per hook its shape, L2 norm and 2048 seeded
sample positions with their fp32 values (the weights are regenerated from the seed by oracle.unet_ref.synth_params)."""
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle import ref_unet as RU, unet_ref as R  # noqa: E402

NS = 2048


def sample_idx(numel, seed):
    g = torch.Generator().manual_seed(seed)
    return torch.randint(0, numel, (min(NS, numel),), generator=g)


def main():
    for tag, base, lat, batch in (("xl", "xl", 16, 2), ("15", "1-5", 16, 1), ("21", "2-1", 8, 2)):
        c0 = 320 if base == "1-5" else 64
        arch = R.tiny_arch(base, time_embed_dim=4 * c0)           # the reference derives time_embed_dim = 4 * block_out_channels[0]
        net = RU.build_reference_unet(arch).eval()
        P = R.synth_params(arch, seed=0)
        net.load_state_dict(P)
        _, prep, _ = RU.reference_unet_class()
        store = prep(base, types.SimpleNamespace(unet=net), None, 1, True)    # accept-all; train_unet=True keeps tensors on the CPU
        I = R.synth_inputs(arch, batch, lat, seed=1, same_prompt=False)
        akw = {"text_embeds": I["text_embeds"], "time_ids": I["time_ids"]} if "text_embeds" in I else None
        with torch.no_grad():
            y = net(I["sample"], I["timestep"][0], I["ctx"], added_cond_kwargs=akw, return_dict=False)[0]
        arrs = {"in:" + k: v.numpy() for k, v in I.items()}
        order = list(store.stored_feats.keys())
        for n, (k, v) in enumerate(store.stored_feats.items()):
            v = v.float().contiguous()
            idx = sample_idx(v.numel(), n)
            arrs["hook:" + k] = v.flatten()[idx].numpy()
            arrs["norm:" + k] = np.float64(v.double().norm().item())
            arrs["shape:" + k] = np.array(v.shape)
        arrs["out"] = y.numpy()
        arrs["meta"] = np.array(repr(dict(base=base, arch=arch, lat=lat, batch=batch, wseed=0, order=order, ns=NS)))
        path = os.path.join(HERE, f"unet_tiny_{tag}.npz")
        np.savez_compressed(path, **arrs)
        print(tag, len(order), "hooks ->", os.path.getsize(path) // 1024, "KiB")


if __name__ == "__main__":
    main()
