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


class LazySynthParams:
    """Seeded synthetic parameters generated ON DEMAND, one tensor at a time, on the GPU (FLUX.1-dev is 11.9 B parameters: a
    materialised fp32 state dict would be 48 GB of host memory next to the 24 GB arena on the device).  `P[name]` returns the fp32
    CPU tensor the oracle multiplies with, `P.device_tensor(name)` the 16-bit device tensor handed to gdf_model_set_param; both
    come from the same generator call, so the two sides see identical values.  Every value is representable in bf16 AND fp16
    (8 significant bits, |w| >= 2^-14 or 0), so one oracle run serves both element types of the MMDiT path."""

    def __init__(self, shapes, is_norm, device="cuda:0", seed=0):
        self.shapes, self.is_norm, self.device, self.seed = shapes, is_norm, device, seed

    def __contains__(self, name):
        return name in self.shapes

    def keys(self):
        return self.shapes.keys()

    def _gen(self, name):
        import zlib
        shape = self.shapes[name]
        g = torch.Generator(device=self.device).manual_seed((zlib.crc32(name.encode()) ^ (self.seed * 0x9E3779B1)) & 0x7FFFFFFF)
        w = torch.randn(shape, generator=g, device=self.device, dtype=torch.float32)
        if self.is_norm(name):
            w = 1.0 + 0.1 * w
        elif name.endswith(".weight"):
            w = w * (shape[1] ** -0.5)
        else:
            w = 0.05 * w
        w = w.to(torch.bfloat16).float()
        w = torch.where(w.abs() < 2.0 ** -14, torch.zeros_like(w), w)
        return w

    def device_tensor(self, name, dtype=torch.bfloat16):
        return self._gen(name).to(dtype)

    def __getitem__(self, name):
        return self._gen(name).cpu()

    def get(self, name, default=None):
        return self[name] if name in self.shapes else default


class _DeviceView:
    """state-dict view of LazySynthParams for _NativeModel.load_state_dict (device tensors of `dtype`, one at a time)"""

    def __init__(self, lazy, dtype):
        self.lazy, self.dtype = lazy, dtype

    def __contains__(self, name):
        return name in self.lazy

    def __getitem__(self, name):
        return self.lazy.device_tensor(name, self.dtype)
