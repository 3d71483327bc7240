"""Golden vectors for the VAE-encoder blocks from the REFERENCE's own modules (build container only).

    python tests/golden/gen_golden_vae.py      # needs /root/reference; writes tests/golden/vae_*.npz

AutoencoderKL itself is un-vendored, but the blocks its encoder is made of are in the reference tree:
ResnetBlock2D(temb_channels=None, eps=1e-6) (feature/diffusers/models/resnet.py), Downsample2D(padding=0)
(feature/diffusers/models/downsampling.py:141-143) and the single-head mid-block Attention with group_norm + residual
(feature/diffusers/models/attention_processor.py Attention + AttnProcessor2_0).  Fixtures are pure data."""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
sys.path.insert(0, HERE)
from oracle import ref_blocks as RB  # noqa: E402
from oracle import vae_ref as VR  # noqa: E402
from gen_golden import init_module, randh, save  # noqa: E402


@torch.no_grad()
def main():
    m = RB.modules()
    g = torch.Generator().manual_seed(21)
    for name, cin, cout in (("vae_resnet_same", 64, 64), ("vae_resnet_shortcut", 64, 128)):
        mod = m.ResnetBlock2D(in_channels=cin, out_channels=cout, temb_channels=None, groups=32, eps=1e-6)
        w = init_module(mod, g)
        x = randh(g, 2, cin, 6, 6, scale=2.0)
        y = mod(x, None)
        save(name, w, {"x": x}, {"y": y}, dict(cin=cin, cout=cout))
        y2 = VR.resnet_block({"r." + k: v for k, v in w.items()}, "r", x)
        assert float((y - y2).abs().max()) < 2e-5
    mod = m.Downsample2D(64, use_conv=True, out_channels=64, padding=0, name="op")
    w = init_module(mod, g)
    x = randh(g, 2, 64, 8, 8)
    y = mod(x)
    save("vae_downsample_pad0", w, {"x": x}, {"y": y}, {})
    assert float((y - VR.downsample_pad0({"d." + k: v for k, v in w.items()}, "d", x)).abs().max()) < 2e-5
    c = 128
    mod = m.Attention(c, heads=1, dim_head=c, rescale_output_factor=1.0, eps=1e-6, norm_num_groups=32,
                      residual_connection=True, bias=True, upcast_softmax=True, _from_deprecated_attn_block=True,
                      processor=m.AttnProcessor2_0())
    w = init_module(mod, g)
    x = randh(g, 2, c, 4, 4, scale=2.0)
    y = mod(x)
    save("vae_mid_attention", w, {"x": x}, {"y": y}, dict(c=c))
    assert float((y - VR.mid_attention({"a." + k: v for k, v in w.items()}, "a", x)).abs().max()) < 2e-5
    print("oracle/vae_ref.py block functions match the reference modules")


if __name__ == "__main__":
    if not RB.available():
        sys.exit("reference tree not found; goldens can only be generated in the build container")
    main()
