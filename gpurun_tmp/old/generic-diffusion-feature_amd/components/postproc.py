"""Output-stage post-processing on the GPU through libgdf.so (include/gdf_ops.h, csrc/post.hip) — the steps right after
the hot path:

  resize_concat      extract_feature.py:113-125   `--aggregate_output`: nearest-resize every layer to the largest H, W, concat
  avg_pool           components/feature_extractor.py:51-53   `feature_resize` (adaptive_avg_pool2d to (H/r, W/r))
  aggregate_maps     components/attention.py:238-244, 141-161 + diffusion_feature.py:492-500   aggregated `attn` feature

Device tensors go through the HIP kernels (and fail loudly if the library is missing); host tensors (accept-all mode hands
out `.cpu()` copies, feature_extractor.py:65-66) are plain data already off the GPU path and use torch on the CPU.
"""
import ctypes as C
import math

import torch
import torch.nn.functional as F

from . import native

_SIG = {
    "gdf_op_resize_concat": (C.c_int, [C.c_void_p, C.c_int, C.c_long, C.c_long, C.c_long, C.c_long, C.c_int, C.c_int, C.c_int, C.c_int,
                                       C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "gdf_op_avg_pool": (C.c_int, [C.c_void_p, C.c_long, C.c_long, C.c_long, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p,
                                  C.c_void_p]),
    "gdf_op_maps_mean": (C.c_int, [C.POINTER(C.c_void_p), C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
}
_lib = None


def _L():
    global _lib
    if _lib is None:
        lib = native.load_library()
        for n, (r, a) in _SIG.items():
            f = getattr(lib, n)
            f.restype, f.argtypes = r, a
        _lib = lib
    return _lib


def _stream(dev):
    return C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)


def resize_concat(feats, size=None):
    """[(B, C_l, H_l, W_l) fp16 / fp32 tensors, any strides] -> (B, sum C_l, S, S) fp16, S = max W_l unless given:
    F.interpolate(v, S) (nearest) of every tensor + torch.cat(dim=1)."""
    feats = list(feats)
    S = int(size or max(v.shape[-1] for v in feats))
    if not feats[0].is_cuda:
        return torch.cat([F.interpolate(v, S) for v in feats], dim=1)
    dev = feats[0].device
    B = feats[0].shape[0]
    ctot = sum(v.shape[1] for v in feats)
    with torch.cuda.device(dev):
        out = torch.empty(B, ctot, S, S, dtype=torch.float16, device=dev)
        off = 0
        for v in feats:
            if v.dtype not in (torch.float16, torch.float32):
                v = v.float()
            b, c, h, w = v.shape
            sb, sc, sy, sx = v.stride()
            native._check(_L().gdf_op_resize_concat(C.c_void_p(v.data_ptr()), int(v.dtype == torch.float32), sb, sc, sy, sx, b, c, h, w,
                                                    C.c_void_p(out.data_ptr()), ctot, off, S, _stream(dev)), "resize_concat")
            off += c
    return out


def avg_pool(feat, r):
    """(B, C, H, W) fp16 -> (B, C, H/r, W/r) fp16 (channels-last storage for device tensors), mean over r x r windows."""
    if r <= 1:
        return feat
    tgt = (feat.shape[2] // r, feat.shape[3] // r)
    b, c, h, w = feat.shape
    if not feat.is_cuda or feat.dtype != torch.float16 or feat.stride(1) != 1 or c % 8 or any(s % 8 for s in feat.stride()[0:1] + feat.stride()[2:]):
        if feat.is_cuda and feat.dtype == torch.float16 and c % 8 == 0:
            return avg_pool(feat.permute(0, 2, 3, 1).contiguous().permute(0, 3, 1, 2), r)   # make it channels-last, then the kernel
        return F.adaptive_avg_pool2d(feat.float(), tgt).to(torch.float16)
    dev = feat.device
    with torch.cuda.device(dev):
        out = torch.empty(b, tgt[0], tgt[1], c, dtype=torch.float16, device=dev)
        native._check(_L().gdf_op_avg_pool(C.c_void_p(feat.data_ptr()), feat.stride(0), feat.stride(2), feat.stride(3), b, c, h, w, r,
                                           C.c_void_p(out.data_ptr()), _stream(dev)), "avg_pool")
    return out.permute(0, 3, 1, 2)


def aggregate_maps(maps_by_category, out_size):
    """{category: [(B, heads, Q, K) fp16 maps in execution order]} -> (B, sum K, out_size, out_size) fp16: head mean, mean over
    the maps of one (category, query-grid size) group, nearest resize, concat in category then first-seen-size order."""
    groups = []
    for cat, maps in maps_by_category.items():
        by_size = {}
        for m in maps:
            by_size.setdefault(int(math.sqrt(m.shape[2])), []).append(m)
        groups.extend(by_size.items())
    first = groups[0][1][0]
    if not first.is_cuda:
        outs = []
        for size, lst in groups:
            acc = [m.float().mean(1).to(torch.float16).float() for m in lst]
            b, q, k = acc[0].shape
            avg = torch.stack(acc).mean(0).reshape(b, size, q // size, k).permute(0, 3, 1, 2)
            outs.append(F.interpolate(avg, size=(out_size, out_size)).to(torch.float16))
        return torch.cat(outs, dim=-3)
    dev = first.device
    B = first.shape[0]
    ktot = sum(lst[0].shape[3] for _, lst in groups)
    with torch.cuda.device(dev):
        out = torch.empty(B, ktot, out_size, out_size, dtype=torch.float16, device=dev)
        off = 0
        for size, lst in groups:
            assert len(lst) <= 32, "more than 32 maps in one (category, size) group"     # SDXL up_* @ 32x32: 3 x 10 blocks = 30; PixArt: 28
            lst = [m.contiguous() for m in lst]
            _, heads, Q, K = lst[0].shape
            mean = torch.empty(B, Q, K, dtype=torch.float32, device=dev)
            ptrs = (C.c_void_p * len(lst))(*[m.data_ptr() for m in lst])
            native._check(_L().gdf_op_maps_mean(ptrs, len(lst), B, heads, Q, K, C.c_void_p(mean.data_ptr()), _stream(dev)), "maps_mean")
            # (B, Q, K) == channels-last image of the logical (B, C = K, size, Q / size) tensor
            wq = Q // size
            native._check(_L().gdf_op_resize_concat(C.c_void_p(mean.data_ptr()), 1, Q * K, 1, wq * K, K, B, K, size, wq,
                                                    C.c_void_p(out.data_ptr()), ktot, off, out_size, _stream(dev)), "resize_concat")
            off += K
            del mean
    return out
