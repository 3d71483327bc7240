"""ctypes binding of include/gdf_ops.h for the kernel-level GPU tests."""
import ctypes as C
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "generic-diffusion-feature_amd"))
from components import native  # noqa: E402

vp, ci, fp = C.c_void_p, C.c_int, C.c_float
OPS = {
    "gdf_op_gemm": (ci, [vp, ci, vp, vp, vp, vp, ci, vp, ci, vp, ci, ci, ci, ci, ci, vp]),
    "gdf_op_conv3x3": (ci, [vp, ci, ci, ci, ci, ci, vp, ci, vp, vp, ci, ci, vp, vp, vp, vp, ci, vp]),
    "gdf_op_conv_in": (ci, [vp, ci, ci, ci, ci, vp, vp, ci, vp, vp, vp]),
    "gdf_op_gemm_split": (ci, [vp, ci, ci, vp, vp, vp, ci, vp, ci, ci, vp, ci, ci, ci, ci, ci, vp]),
    "gdf_op_conv3x3_split": (ci, [vp, ci, ci, ci, ci, ci, ci, vp, ci, vp, ci, ci, vp, vp, ci, ci, vp, vp]),
    "gdf_op_layernorm_split": (ci, [vp, ci, ci, ci, fp, vp, vp, vp, ci, ci, vp]),
    "gdf_op_groupnorm_split": (ci, [vp, ci, vp, ci, ci, ci, ci, ci, fp, vp, vp, ci, vp, ci, ci, vp, vp]),
    "gdf_op_attention_split": (ci, [vp, ci, vp, ci, vp, ci, vp, ci, ci, ci, ci, ci, ci, ci, vp, vp]),
    "gdf_op_attention_pair": (ci, [vp, ci, vp, ci, vp, ci, ci, vp, ci, ci, ci, ci, ci, ci, ci, vp]),
    "gdf_op_attention": (ci, [vp, ci, vp, ci, vp, ci, vp, ci, ci, ci, ci, ci, ci, vp, vp]),
    "gdf_op_groupnorm_scratch_bytes": (C.c_size_t, [ci, ci, ci]),
    "gdf_op_groupnorm": (ci, [vp, vp, ci, ci, ci, ci, ci, fp, vp, vp, ci, vp, vp, vp]),
    "gdf_op_layernorm": (ci, [vp, vp, ci, ci, ci, fp, vp, vp, vp, vp]),
    "gdf_op_copy2d": (ci, [vp, vp, ci, vp, ci, ci, ci, vp]),
    "gdf_op_relayout_conv3": (ci, [vp, vp, ci, ci, vp]),
    "gdf_op_relayout_geglu": (ci, [vp, vp, vp, vp, ci, ci, ci, vp]),
    "gdf_op_sincos_pos_embed": (ci, [vp, ci, ci, ci, ci, fp, vp]),
    "gdf_op_softmax_rows": (ci, [vp, ci, ci, ci, fp, vp]),
    "gdf_op_small_linear": (ci, [vp, ci, ci, ci, vp, vp, ci, ci, ci, vp, ci, vp]),
    "gdf_op_set_e16": (ci, [ci]),
    "gdf_op_gemm_dit": (ci, [vp, ci, vp, vp, ci, vp, ci, ci, ci, ci, ci, vp, ci, vp, ci, vp, ci, vp, ci, ci, ci, ci, ci, vp]),
    "gdf_op_quant_rows_fp8": (ci, [vp, ci, ci, ci, ci, vp, ci, vp, vp]),
    "gdf_op_gemm_mx": (ci, [vp, ci, vp, vp, vp, vp, ci, vp, ci, vp, ci, vp, ci, ci, ci, ci, vp]),
    "gdf_op_layernorm_mod": (ci, [vp, ci, ci, ci, fp, vp, vp, ci, ci, ci, ci, vp, vp]),
    "gdf_op_qk_norm_rope": (ci, [vp, ci, ci, ci, ci, ci, vp, vp, fp, vp, vp, ci, ci, vp]),
    "gdf_op_rope_table": (ci, [vp, ci, ci, ci, ci, vp, vp, ci, vp]),
    "gdf_op_attention_joint": (ci, [vp, ci, vp, ci, vp, ci, vp, ci, ci, ci, ci, ci, ci, vp]),
}


def lib():
    L = native.load_library()
    for n, (r, a) in OPS.items():
        f = getattr(L, n)
        f.restype, f.argtypes = r, a
    return L


def P(t):
    return vp(t.data_ptr()) if t is not None else vp(0)


def stream():
    return vp(torch.cuda.current_stream().cuda_stream)


def ok(rc, L):
    assert rc == 0, L.gdf_last_error().decode()


def rel(a, b):
    a, b = a.float().cpu(), b.float().cpu()
    return float((a - b).norm() / (b.norm() + 1e-30))
