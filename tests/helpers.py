import torch

from oracle import unet_ref as R


def cfg_from_oracle_arch(a):
    tt = bool(a["addition_embed"])
    return dict(in_channels=4, out_channels=4, block_out_channels=tuple(a["block_out_channels"]),
                has_attn=tuple(int(x) for x in a["down_attn"]), transformer_layers=tuple(a["transformer_layers"]),
                heads=tuple(a["heads"]), layers_per_block=a["layers_per_block"], cross_attention_dim=a["cross_dim"],
                use_linear_projection=int(a["linear_proj"]), time_embed_dim=a["time_embed_dim"],
                addition_embed_text_time=int(tt), addition_time_embed_dim=a.get("addition_time_embed_dim", 0) if tt else 0,
                add_in_dim=a.get("add_in_dim", 0) if tt else 0)


def oracle_run(arch, P, I, ids=None):
    st = R.Store({k: True for k in ids} if ids else None)
    R.unet_forward(P, arch, I["sample"], I["timestep"], I["ctx"], I.get("text_embeds"), I.get("time_ids"), store=st)
    return st.feats


def rel_l2(got, ref):
    got, ref = got.float().cpu(), ref.float().cpu()
    return float((got - ref).norm() / (ref.norm() + 1e-30))
