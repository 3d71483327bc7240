"""CPU ORACLE helper (test infrastructure only): the fp16-OPERAND FLOOR of the hot path.

`fp16_operands()` is a context manager under which the fp32 oracles (oracle/unet_ref.py, ...) round every
matrix-multiply operand to fp16 before multiplying — and nothing else: accumulation, normalisation statistics,
softmax, activations and the residual stream stay fp32.  That is the arithmetic ANY implementation with fp16
MFMA operands performs at best (the reference's own fp16 GPU path rounds more: its residual stream and norm
outputs are fp16 too), so the error of such a run against the plain fp32 oracle is the floor below which no
fp16-operand kernel can go.  The full-size GPU parity tests assert the HIP path against the north-star bound AND
against this floor (tests/test_gpu_fullsize.py); tools/operand_floor.py prints the per-kind table quoted in DESIGN.md.

What is rounded (= what libgdf.so stores / stages as fp16):
  * inputs of F.linear / F.conv2d (LayerNorm / GroupNorm outputs, q, k, v, attention output, GEGLU inner, fp16 shadow
    of the stream where a conv / projection reads it), weights are fp16-exact already;
  * inputs of F.group_norm (the kernels read the fp16 image of the tensor; LayerNorm reads the fp32 master);
  * q, k, v and the probabilities P of attention (P feeds the PV MFMA as fp16).
Time-embedding vectors stay fp32 (small_linear_kernel works on fp32 vectors).
"""
import contextlib

import torch
import torch.nn.functional as F


def _r(x):
    return x.to(torch.float16).to(torch.float32) if x.dtype == torch.float32 and x.dim() >= 2 else x


@contextlib.contextmanager
def fp16_operands(vec_rows=64):
    """2-D linear inputs with at most `vec_rows` rows are left alone (the (B, C) time-embedding vectors)."""
    lin, conv, gn, sdpa, mm, sm = F.linear, F.conv2d, F.group_norm, F.scaled_dot_product_attention, torch.matmul, torch.softmax

    def linear(x, w, b=None):
        if x.dim() == 2 and x.shape[0] <= vec_rows:
            return lin(x, w, b)                       # (B, C) embedding vectors: fp32 on the GPU as well
        return lin(_r(x), w, b)

    def conv2d(x, w, b=None, *a, **k):
        return conv(_r(x), w, b, *a, **k)

    def group_norm(x, *a, **k):
        return gn(_r(x), *a, **k)

    def matmul(a, b):
        return mm(_r(a), _r(b))

    def attention(q, k, v, *a, **kw):
        q, k, v = _r(q), _r(k), _r(v)
        scale = kw.get("scale") or q.shape[-1] ** -0.5
        p = sm(mm(q, k.transpose(-1, -2)) * scale, dim=-1)
        return mm(_r(p), v)

    F.linear, F.conv2d, F.group_norm, F.scaled_dot_product_attention, torch.matmul = linear, conv2d, group_norm, attention, matmul
    try:
        yield
    finally:
        F.linear, F.conv2d, F.group_norm, F.scaled_dot_product_attention, torch.matmul = lin, conv, gn, sdpa, mm


def kind_of(hook_id):
    """Hook kind used for per-kind tolerances: last token, `-out` ids keep their qualifier (res-out, vit-out, block-out)."""
    p = hook_id.split("-")
    if p[-1] == "out":
        q = p[-2]
        return "block-out" if q.startswith("block") else q + "-out"
    return {"inner": "ffn-inner", "increment": "res-increment"}.get(p[-1], p[-1])
