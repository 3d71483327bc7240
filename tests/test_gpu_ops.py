"""Kernel-level parity (-m gpu): each HIP kernel of libgdf.so, called through the C ABI (include/gdf_ops.h),
against a plain PyTorch fp32 CPU computation of the same op on the same fp16-rounded inputs.
Tolerances: fp16 storage of outputs => relative L2 error <= 1e-3 per tensor (fp32 outputs: 2e-4)."""
import math
import os

import pytest
import torch
import torch.nn.functional as F

from ops_binding import P, lib, ok, rel, stream

pytestmark = pytest.mark.gpu
TOL16, TOL32 = 1e-3, 2e-4


def rnd(*shape, scale=1.0, seed=0):
    g = torch.Generator().manual_seed(seed + sum(shape))
    return (torch.randn(*shape, generator=g) * scale).half()


@pytest.mark.parametrize("variant", [0, 128, 160, 256, 320, 932])     # 932: 8-phase main loop on the 256x320 tile
@pytest.mark.parametrize("M,N,K", [(128, 128, 64), (256, 320, 320), (200, 72, 128), (1232, 640, 2048), (64, 1280, 1280)])
def test_gemm_bias_residual(M, N, K, variant):
    L = lib()
    A, W = rnd(M, K), rnd(N, K, scale=K ** -0.5)
    bias, res = rnd(N).float(), rnd(M, N).float()
    ref = A.float() @ W.float().t() + bias + res
    Ad, Wd, bd, rd = A.cuda(), W.cuda(), bias.cuda(), res.cuda()
    o16 = torch.zeros(M, N, dtype=torch.half, device="cuda"); o32 = torch.zeros(M, N, device="cuda")
    ok(L.gdf_op_gemm(P(Ad), K, P(Wd), P(bd), P(rd), None, N, P(o16), N, P(o32), N, M, N, K, variant << 8, stream()), L)
    torch.cuda.synchronize()
    assert rel(o32, ref) < TOL32 and rel(o16, ref) < TOL16
    # fp16 residual + strided A / out (leading dimensions larger than the logical width)
    Abig = torch.zeros(M, K + 64, dtype=torch.half); Abig[:, 32:32 + K] = A
    obig = torch.zeros(M, N + 24, dtype=torch.half, device="cuda")
    r16 = res.half().cuda()
    Ab = Abig.cuda()
    ok(L.gdf_op_gemm(C_off(Ab, 32), K + 64, P(Wd), P(bd), None, P(r16), N, C_off(obig, 8), N + 24, None, 0, M, N, K,
                     variant << 8, stream()), L)
    torch.cuda.synchronize()
    ref2 = A.float() @ W.float().t() + bias + res.half().float()
    assert rel(obig[:, 8:8 + N], ref2) < TOL16
    assert float(obig[:, :8].abs().max()) == 0 and float(obig[:, 8 + N:].abs().max()) == 0


def C_off(t, cols):
    import ctypes
    return ctypes.c_void_p(t.data_ptr() + cols * t.element_size())


def test_gemm_asymmetric_identity():
    """A = I against an asymmetric B catches row/col swaps in the MFMA C layout."""
    L = lib()
    M = N = K = 128
    A = torch.eye(M).half()
    W = (torch.arange(N)[:, None] * 0.01 + torch.arange(K)[None, :] * 0.003).half()
    o32 = torch.zeros(M, N, device="cuda")
    Ad, Wd = A.cuda(), W.cuda()
    ok(L.gdf_op_gemm(P(Ad), K, P(Wd), None, None, None, 0, None, 0, P(o32), N, M, N, K, 0, stream()), L)
    torch.cuda.synchronize()
    assert torch.allclose(o32.cpu(), W.float().t(), atol=1e-3)


@pytest.mark.parametrize("M,C,variant", [(128, 64, 0), (520, 320, 128), (520, 320, 256), (520, 320, 320), (300, 640, 0),
                                          (520, 320, 825), (1100, 640, 825),
                                          (4200, 640, 825)])       # 17 x 20 = 340 tiles: persistent launch, 84 workgroups take a second tile
def test_gemm_geglu(M, C, variant):
    group = 16
    L = lib()
    x, W, b = rnd(M, C), rnd(8 * C, C, scale=C ** -0.5), rnd(8 * C).float()
    hg = x.float() @ W.float().t() + b
    h, g = hg.chunk(2, -1)
    ref = h * F.gelu(g)
    Wd = torch.empty_like(W, device="cuda"); bd = torch.empty(8 * C, device="cuda")
    Ws, bs, xd = W.cuda(), b.cuda(), x.cuda()
    ok(L.gdf_op_relayout_geglu(P(Ws), P(bs), P(Wd), P(bd), 8 * C, C, group, stream()), L)
    out = torch.zeros(M, 4 * C, dtype=torch.half, device="cuda")
    ok(L.gdf_op_gemm(P(xd), C, P(Wd), P(bd), None, None, 0, P(out), 4 * C, None, 0, M, 8 * C, C, 1 | (variant << 8), stream()), L)
    torch.cuda.synchronize()
    assert rel(out, ref) < TOL16


@pytest.mark.parametrize("variant", [0, 128, 160, 256, 320, 932, 826])     # 932 / 826: 8-phase main loops, conv A operand
@pytest.mark.parametrize("B,H,W,Cin,Cout,stride,ups", [
    (2, 8, 8, 64, 64, 1, 0), (1, 12, 10, 128, 192, 1, 0), (2, 8, 8, 64, 128, 2, 0), (2, 6, 6, 64, 64, 1, 1),
    (1, 16, 16, 320, 320, 1, 0), (2, 20, 12, 128, 320, 1, 0), (1, 10, 10, 192, 640, 1, 1), (2, 16, 16, 64, 320, 2, 0),
    (2, 20, 12, 128, 256, 1, 0), (1, 24, 24, 64, 512, 1, 0)])
def test_conv3x3(B, H, W, Cin, Cout, stride, ups, variant):
    L = lib()
    x = rnd(B, Cin, H, W); w = rnd(Cout, Cin, 3, 3, scale=(9 * Cin) ** -0.5)
    bias, temb = rnd(Cout).float(), rnd(B, Cout).float()
    xi = F.interpolate(x.float(), scale_factor=2.0, mode="nearest") if ups else x.float()
    inc = F.conv2d(xi, w.float(), bias, stride=stride, padding=1) + temb[:, :, None, None]
    OH, OW = inc.shape[2], inc.shape[3]
    res = rnd(B, Cout, OH, OW, seed=5).float()
    ref = inc + res
    x_nhwc = x.permute(0, 2, 3, 1).contiguous().cuda()
    wd = torch.empty(Cout, 9 * Cin, dtype=torch.half, device="cuda")
    ws, bd, td = w.cuda(), bias.cuda(), temb.cuda()
    ok(L.gdf_op_relayout_conv3(P(ws), P(wd), Cout, Cin, stream()), L)
    res_nhwc = res.permute(0, 2, 3, 1).contiguous().cuda()
    aux = torch.zeros(B, OH, OW, Cout, dtype=torch.half, device="cuda")
    o16 = torch.zeros_like(aux); o32 = torch.zeros(B, OH, OW, Cout, device="cuda")
    ok(L.gdf_op_conv3x3(P(x_nhwc), Cin, B, H, W, Cin, P(wd), Cout, P(bd), P(td), stride, ups,
                        P(res_nhwc), P(aux), P(o16), P(o32), variant << 8, stream()), L)
    torch.cuda.synchronize()
    assert rel(aux.permute(0, 3, 1, 2), inc) < TOL16
    assert rel(o32.permute(0, 3, 1, 2), ref) < TOL32
    assert rel(o16.permute(0, 3, 1, 2), ref) < TOL16


@pytest.mark.parametrize("B,H,W,Cin,Cout,stride,splitk", [(2, 8, 8, 1280, 1280, 1, 0), (3, 8, 8, 640, 256, 1, 4), (2, 12, 10, 320, 384, 2, 3),
                                                        (1, 8, 8, 2560, 1280, 1, 8)])
def test_conv3x3_splitk(B, H, W, Cin, Cout, stride, splitk):
    """Deterministic split-K (include/gdf_ops.h gdf_op_conv3x3_splitk; the plan builder uses it for few-tile, long-K convs —
    SD1.5's 8x8 level): same epilogue semantics as gdf_op_conv3x3, vs fp32 F.conv2d; two launches are bit-identical."""
    import ctypes
    L = lib()
    L.gdf_op_splitk_factor.restype = ctypes.c_int
    x = rnd(B, Cin, H, W); w = rnd(Cout, Cin, 3, 3, scale=(9 * Cin) ** -0.5)
    bias, temb = rnd(Cout).float(), rnd(B, Cout).float()
    inc = F.conv2d(x.float(), w.float(), bias, stride=stride, padding=1) + temb[:, :, None, None]
    OH, OW = inc.shape[2], inc.shape[3]
    res = rnd(B, Cout, OH, OW, seed=5).float()
    ref = inc + res
    M = B * OH * OW
    heur = L.gdf_op_splitk_factor(M, Cout, 9 * Cin, 1)
    if splitk == 0:
        assert heur > 1                                          # the SD1.5 8x8 shape class must take the split path
    S = splitk or heur
    x_nhwc = x.permute(0, 2, 3, 1).contiguous().cuda()
    wd = torch.empty(Cout, 9 * Cin, dtype=torch.half, device="cuda")
    ws_, bd, td = w.cuda(), bias.cuda(), temb.cuda()
    ok(L.gdf_op_relayout_conv3(P(ws_), P(wd), Cout, Cin, stream()), L)
    res_nhwc = res.permute(0, 2, 3, 1).contiguous().cuda()
    work = torch.empty(S * M * Cout, device="cuda")
    outs = []
    for rep in range(2):
        aux = torch.zeros(B, OH, OW, Cout, dtype=torch.half, device="cuda")
        o16 = torch.zeros_like(aux); o32 = torch.zeros(B, OH, OW, Cout, device="cuda")
        ok(L.gdf_op_conv3x3_splitk(P(x_nhwc), Cin, B, H, W, Cin, P(wd), Cout, P(bd), P(td), stride, 0,
                                   P(res_nhwc), P(aux), P(o16), P(o32), splitk, P(work), stream()), L)
        torch.cuda.synchronize()
        outs.append((aux, o16, o32))
    aux, o16, o32 = outs[0]
    assert rel(aux.permute(0, 3, 1, 2), inc) < TOL16
    assert rel(o32.permute(0, 3, 1, 2), ref) < TOL32
    assert rel(o16.permute(0, 3, 1, 2), ref) < TOL16
    assert all(torch.equal(a, b) for a, b in zip(outs[0], outs[1]))                  # fixed summation order: run-to-run bit-identical


def test_conv3x3_narrow_cout4():
    L = lib()
    B, H, W, Cin, Cout = 2, 8, 8, 64, 4
    x = rnd(B, Cin, H, W); w = rnd(Cout, Cin, 3, 3, scale=(9 * Cin) ** -0.5); bias = rnd(Cout).float()
    ref = F.conv2d(x.float(), w.float(), bias, padding=1)
    wd = torch.empty(Cout, 9 * Cin, dtype=torch.half, device="cuda")
    ws, bd, xd = w.cuda(), bias.cuda(), x.permute(0, 2, 3, 1).contiguous().cuda()
    ok(L.gdf_op_relayout_conv3(P(ws), P(wd), Cout, Cin, stream()), L)
    o16 = torch.zeros(B, H, W, Cout, dtype=torch.half, device="cuda")
    ok(L.gdf_op_conv3x3(P(xd), Cin, B, H, W, Cin, P(wd), Cout, P(bd), None,
                        1, 0, None, None, P(o16), None, 1, stream()), L)
    torch.cuda.synchronize()
    assert rel(o16.permute(0, 3, 1, 2), ref) < TOL16


def test_conv_in():
    L = lib()
    B, H, W, Cout = 2, 8, 12, 64
    x = rnd(B, 4, H, W); w = rnd(Cout, 4, 3, 3, scale=1 / 6.0); bias = rnd(Cout).float()
    ref = F.conv2d(x.float(), w.float(), bias, padding=1)
    out = torch.zeros(B, H, W, Cout, dtype=torch.half, device="cuda")
    scratch = torch.zeros(B * H * W * 16 + Cout * 256, dtype=torch.uint8, device="cuda")
    xd, wd, bd = x.cuda(), w.cuda(), bias.cuda()
    ok(L.gdf_op_conv_in(P(xd), B, 4, H, W, P(wd), P(bd), Cout, P(out), P(scratch), stream()), L)
    torch.cuda.synchronize()
    assert rel(out.permute(0, 3, 1, 2), ref) < TOL16


@pytest.mark.parametrize("B,heads,Sq,Sk,D", [(1, 2, 64, 64, 64), (2, 2, 36, 36, 32), (2, 3, 200, 77, 64),
                                              (1, 4, 256, 256, 40), (1, 2, 130, 77, 80), (1, 1, 64, 200, 160),
                                              (1, 2, 1024, 1024, 64),
                                              # map-kernel variants: loader wave at its minimum of two key tiles (D = 32) and with
                                              # ragged batches of steps (D = 40, 6 tiles); FULL without loader wave (Sk % 128 != 0,
                                              # D = 64 / 80)
                                              (1, 2, 128, 128, 32), (2, 2, 384, 384, 40), (2, 2, 384, 192, 40),
                                              (1, 2, 256, 512, 64), (1, 2, 128, 320, 80)])
def test_attention(B, heads, Sq, Sk, D):
    L = lib()
    C = heads * D
    q, k, v = rnd(B, Sq, C), rnd(B, Sk, C, seed=1), rnd(B, Sk, C, seed=2)
    qh = q.float().view(B, Sq, heads, D).transpose(1, 2)
    kh = k.float().view(B, Sk, heads, D).transpose(1, 2)
    vh = v.float().view(B, Sk, heads, D).transpose(1, 2)
    probs = torch.softmax(qh @ kh.transpose(-1, -2) * D ** -0.5, -1)
    ref = (probs @ vh).transpose(1, 2).reshape(B, Sq, C)
    o = torch.zeros(B, Sq, C, dtype=torch.half, device="cuda")
    qd, kd, vd = q.cuda(), k.cuda(), v.cuda()
    ok(L.gdf_op_attention(P(qd), C, P(kd), C, P(vd), C, P(o), C, B, heads, Sq, Sk, D, None, stream()), L)
    torch.cuda.synchronize()
    assert rel(o, ref) < 2e-3
    if Sq * Sk <= 512 * 512:
        o2 = torch.zeros_like(o); mp = torch.zeros(B, heads, Sq, Sk, dtype=torch.half, device="cuda")
        ok(L.gdf_op_attention(P(qd), C, P(kd), C, P(vd), C, P(o2), C, B, heads, Sq, Sk, D, P(mp), stream()), L)
        torch.cuda.synchronize()
        assert rel(mp, probs) < TOL16 and rel(o2, ref) < TOL16


def test_attention_forced_rescale():
    """A key that spikes late in the sequence forces the online-softmax rescale branch."""
    L = lib()
    B, heads, S, D = 1, 1, 256, 64
    q, k, v = rnd(B, S, D), rnd(B, S, D, seed=1), rnd(B, S, D, seed=2)
    k[0, 200] = q[0, 7] * 4.0
    s = (q.float() @ k.float().transpose(-1, -2)) * D ** -0.5
    ref = torch.softmax(s, -1) @ v.float()
    o = torch.zeros(B, S, D, dtype=torch.half, device="cuda")
    qd, kd, vd = q.cuda(), k.cuda(), v.cuda()
    ok(L.gdf_op_attention(P(qd), D, P(kd), D, P(vd), D, P(o), D, B, heads, S, S, D, None, stream()), L)
    torch.cuda.synchronize()
    assert rel(o, ref) < 2e-3 and rel(o[0, 7], ref[0, 7]) < 2e-3


@pytest.mark.parametrize("B,HW,C,ld,silu,eps,f32", [(2, 64, 64, 64, 1, 1e-5, 0), (2, 300, 320, 384, 1, 1e-5, 0),
                                                    (1, 1024, 2560, 2560, 0, 1e-6, 0), (2, 64, 128, 128, 1, 1e-5, 1),
                                                    # single-launch kernel (<= 32x32 pixels, >= 64 slabs): 10 / 30 / 40 / 80
                                                    # channels per group, strided input, fp32 input, ragged row count
                                                    (16, 1024, 1280, 1280, 1, 1e-5, 0), (8, 256, 960, 1280, 1, 1e-5, 0),
                                                    (32, 64, 320, 320, 0, 1e-6, 0), (4, 1000, 2560, 2560, 1, 1e-5, 1),
                                                    (16, 100, 1920, 1920, 1, 1e-5, 0)])
def test_groupnorm(B, HW, C, ld, silu, eps, f32):
    L = lib()
    x = rnd(B, HW, ld, scale=2.0) + 0.5
    gamma, beta = (1 + 0.1 * rnd(C).float()), 0.1 * rnd(C, seed=3).float()
    xr = x[:, :, :C].float().permute(0, 2, 1)
    ref = F.group_norm(xr, 32, gamma, beta, eps)
    if silu:
        ref = F.silu(ref)
    ref = ref.permute(0, 2, 1)
    y = torch.zeros(B, HW, C, dtype=torch.half, device="cuda")
    scratch = torch.zeros(L.gdf_op_groupnorm_scratch_bytes(B, HW, C) + 1024, dtype=torch.uint8, device="cuda")
    xd = x.float().cuda() if f32 else x.cuda()
    gd, bd = gamma.cuda(), beta.cuda()
    ok(L.gdf_op_groupnorm(None if f32 else P(xd), P(xd) if f32 else None, ld, B, HW, C, 32, eps, P(gd),
                          P(bd), silu, P(y), P(scratch), stream()), L)
    torch.cuda.synchronize()
    assert rel(y, ref) < TOL16


@pytest.mark.parametrize("R,C,f32", [(100, 64, 0), (257, 320, 1), (64, 640, 0), (33, 1280, 1)])
def test_layernorm(R, C, f32):
    L = lib()
    x = rnd(R, C, scale=3.0) + 1.0
    gamma, beta = (1 + 0.1 * rnd(C).float()), 0.1 * rnd(C, seed=3).float()
    ref = F.layer_norm(x.float(), (C,), gamma, beta, 1e-5)
    y = torch.zeros(R, C, dtype=torch.half, device="cuda")
    xd = x.float().cuda() if f32 else x.cuda()
    gd, bd = gamma.cuda(), beta.cuda()
    ok(L.gdf_op_layernorm(None if f32 else P(xd), P(xd) if f32 else None, C, R, C, 1e-5, P(gd), P(bd),
                          P(y), stream()), L)
    torch.cuda.synchronize()
    assert rel(y, ref) < TOL16


def test_copy2d_hook_store():
    L = lib()
    src = rnd(50, 96)
    dst = torch.zeros(50, 40, dtype=torch.half, device="cuda")
    sd = src.cuda()
    ok(L.gdf_op_copy2d(C_off(sd, 16), None, 96, P(dst), 40, 50, 40, stream()), L)
    s4 = rnd(37, 4)
    s4d = s4.cuda()
    d4 = torch.zeros(37, 4, dtype=torch.half, device="cuda")
    ok(L.gdf_op_copy2d(P(s4d), None, 4, P(d4), 4, 37, 4, stream()), L)
    torch.cuda.synchronize()
    assert torch.equal(dst.cpu(), src[:, 16:56]) and torch.equal(d4.cpu(), s4)      # bit-exact: pure byte movement


def test_gemm_8phase_reproduces_ring_bitwise():
    """Race screen (tools/stress_gemm8.py): the 8-phase main loops keep the K order of the LDS-ring kernels, so every launch must
    match them bit for bit, also with a bandwidth hog on a second stream."""
    import subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "stress_gemm8.py"), "6"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "0 mismatches" in r.stdout


def test_postproc_kernels_vs_torch():
    """csrc/post.hip through components/postproc.py: resize_concat (nearest + concat, extract_feature.py:113-125), avg_pool
    (`feature_resize`, feature_extractor.py:51-53), aggregate_maps (AttentionStore, attention.py:141-161) vs ATen / golden."""
    import torch.nn.functional as F
    from components.postproc import aggregate_maps, avg_pool, resize_concat
    g = torch.Generator(device="cuda").manual_seed(0)
    cl = lambda t: t.permute(0, 2, 3, 1).contiguous().permute(0, 3, 1, 2)          # channels-last storage, like every hook
    feats = [cl(torch.randn(3, 70, 12, 12, device="cuda", generator=g).half()), torch.randn(3, 64, 160, 160, device="cuda", generator=g).half(),
             cl(torch.randn(3, 8, 40, 40, device="cuda", generator=g)), cl(torch.randn(3, 130, 53, 53, device="cuda", generator=g).half())]
    got = resize_concat(feats)
    ref = torch.cat([F.interpolate(v.float(), 160).half() for v in feats], dim=1)
    assert got.shape == (3, 272, 160, 160) and torch.equal(got, ref)                # byte-exact (incl. non-integer ratios 53 -> 160)
    x = cl(torch.randn(2, 320, 32, 48, device="cuda", generator=g).half())
    for r in (2, 4):
        p = avg_pool(x, r)
        assert p.shape == (2, 320, 32 // r, 48 // r)
        assert torch.allclose(p.float(), F.adaptive_avg_pool2d(x.float(), (32 // r, 48 // r)), atol=2e-3)
    from test_host_cpu import _attn_golden
    from oracle import attn_agg_ref as AR
    meta, maps, cases = _attn_golden()
    for sel, want in cases:
        by_cat = {c: [m.cuda() for h, m in maps if AR.category_of(h) == c and meta["min_size"] ** 2 <= m.shape[2] <= meta["max_size"] ** 2]
                  for c in sel}
        out = aggregate_maps(by_cat, meta["out_size"])
        assert out.is_cuda and out.dtype == torch.float16 and torch.allclose(out.float().cpu(), want, atol=1e-3)


# ---- split fp16 hi + lo operands (the opt-in "precise" plans): kernel-level checks through include/gdf_ops.h ----
def _split(x32):
    hi = x32.half()
    lo = (x32 - hi.float()).half()
    return hi, lo


@pytest.mark.parametrize("M,N,K", [(256, 320, 320), (200, 640, 1280), (1232, 1280, 640), (300, 160, 2560)])
def test_gemm_split_operands(M, N, K):
    """A = [hi | lo] (lo K + 32 columns further: a gap like the skip-concat buffers have), contraction over 2K against W read twice.
    Against fp64 of (hi + lo) W^T: fp32-accumulation accuracy (< 3e-6), where the plain fp16-operand GEMM of the same data is at the
    operand rounding (~2.9e-4); the output pair hi + lo reproduces the fp32 result to 2^-21."""
    L = lib()
    g = torch.Generator().manual_seed(M + N + K)
    a32 = torch.randn(M, K, generator=g)
    W = (torch.randn(N, K, generator=g) * K ** -0.5).half()
    bias, res = torch.randn(N, generator=g), torch.randn(M, N, generator=g)
    hi, lo = _split(a32)
    a_lo = K + 32
    A = torch.zeros(M, a_lo + K + 8, dtype=torch.half)
    A[:, :K] = hi; A[:, a_lo:a_lo + K] = lo
    ref = ((hi.double() + lo.double()) @ W.double().t() + bias.double() + res.double())
    Ad, Wd, bd, rd = A.cuda(), W.cuda(), bias.cuda(), res.cuda()
    o_lo = N + 8
    o16 = torch.zeros(M, o_lo + N, dtype=torch.half, device="cuda"); o32 = torch.zeros(M, N, device="cuda")
    ok(L.gdf_op_gemm_split(P(Ad), A.shape[1], a_lo, P(Wd), P(bd), P(rd), N, P(o16), o16.shape[1], o_lo, P(o32), N, M, N, K, 0, stream()), L)
    torch.cuda.synchronize()
    e32 = float((o32.double().cpu() - ref).norm() / ref.norm())
    assert e32 < 3e-6, e32
    pair = o16[:, :N].double().cpu() + o16[:, o_lo:o_lo + N].double().cpu()
    assert float((pair - ref).norm() / ref.norm()) < 3e-6
    assert torch.equal(o16[:, :N].cpu(), o32.cpu().half())                       # hi = fp16(v): what a hook of this tensor stores
    # the plain fp16-operand GEMM of the same activation, for scale
    o32p = torch.zeros(M, N, device="cuda")
    hid = hi.cuda()
    ok(L.gdf_op_gemm(P(hid), K, P(Wd), P(bd), P(rd), None, N, None, 0, P(o32p), N, M, N, K, 0, stream()), L)
    torch.cuda.synchronize()
    ref_true = a32.double() @ W.double().t() + bias.double() + res.double()
    ep = float((o32p.double().cpu() - ref_true).norm() / ref_true.norm())
    es = float((o32.double().cpu() - ref_true).norm() / ref_true.norm())
    assert es < 0.02 * ep, (es, ep)


def test_gemm_split_geglu():
    L = lib()
    M, C = 512, 320
    g = torch.Generator().manual_seed(3)
    a32 = torch.randn(M, C, generator=g)
    W = (torch.randn(8 * C, C, generator=g) * C ** -0.5).half(); b = torch.randn(8 * C, generator=g) * 0.1
    hi, lo = _split(a32)
    A = torch.cat([hi, lo], 1).cuda()
    Wi = torch.empty_like(W).cuda(); bi = torch.empty(8 * C, device="cuda")
    Wd, bd = W.cuda(), b.cuda()
    ok(L.gdf_op_relayout_geglu(P(Wd), P(bd), P(Wi), P(bi), 8 * C, C, 16, stream()), L)
    y = (hi.double() + lo.double()) @ W.double().t() + b.double()
    h, gate = y.chunk(2, -1)
    ref = h * F.gelu(gate)
    o16 = torch.zeros(M, 8 * C, dtype=torch.half, device="cuda")                # [hi(4C) | lo(4C)]
    ok(L.gdf_op_gemm_split(P(A), 2 * C, C, P(Wi), P(bi), None, 0, P(o16), 8 * C, 4 * C, None, 0, M, 8 * C, C, 1, stream()), L)
    torch.cuda.synchronize()
    pair = o16[:, :4 * C].double().cpu() + o16[:, 4 * C:].double().cpu()
    e = float((pair - ref).norm() / ref.norm())
    assert e < 2e-5, e                                                          # the single-exp GELU's own error (5e-6 relative)


@pytest.mark.parametrize("B,H,W,Cin,Cout,stride,ups", [(2, 8, 8, 64, 64, 1, 0), (1, 16, 16, 320, 320, 1, 0), (2, 12, 10, 128, 320, 2, 0),
                                                        (1, 10, 10, 192, 640, 1, 1), (2, 32, 32, 320, 320, 1, 0)])
def test_conv3x3_split_operands(B, H, W, Cin, Cout, stride, ups):
    L = lib()
    g = torch.Generator().manual_seed(B + H + Cin + Cout)
    x32 = torch.randn(B, Cin, H, W, generator=g)
    w = (torch.randn(Cout, Cin, 3, 3, generator=g) * (9 * Cin) ** -0.5).half()
    bias = torch.randn(Cout, generator=g)
    hi, lo = _split(x32)
    xs = hi.double() + lo.double()
    xi = F.interpolate(xs, scale_factor=2.0, mode="nearest") if ups else xs
    ref = F.conv2d(xi, w.double(), bias.double(), stride=stride, padding=1)
    OH, OW = ref.shape[2], ref.shape[3]
    a_lo = Cin + 64                                                              # a gap between the halves (concat buffers)
    xn = torch.zeros(B, H, W, a_lo + Cin, dtype=torch.half)
    xn[..., :Cin] = hi.permute(0, 2, 3, 1); xn[..., a_lo:] = lo.permute(0, 2, 3, 1)
    xd, ws, bd = xn.cuda(), w.cuda(), bias.cuda()
    wd = torch.empty(Cout, 9 * Cin, dtype=torch.half, device="cuda")
    ok(L.gdf_op_relayout_conv3(P(ws), P(wd), Cout, Cin, stream()), L)
    o16 = torch.zeros(B, OH, OW, 2 * Cout, dtype=torch.half, device="cuda"); o32 = torch.zeros(B, OH, OW, Cout, device="cuda")
    ok(L.gdf_op_conv3x3_split(P(xd), xn.shape[-1], a_lo, B, H, W, Cin, P(wd), Cout, P(bd), stride, ups, None, P(o16), 2 * Cout, Cout,
                              P(o32), stream()), L)
    torch.cuda.synchronize()
    e = float((o32.permute(0, 3, 1, 2).double().cpu() - ref).norm() / ref.norm())
    assert e < 3e-6, e
    pair = (o16[..., :Cout].double() + o16[..., Cout:].double()).permute(0, 3, 1, 2).cpu()
    assert float((pair - ref).norm() / ref.norm()) < 3e-6


def test_norms_and_attention_split_outputs():
    """LayerNorm / GroupNorm (+SiLU) / attention outputs as split pairs: hi + lo reproduces the fp32 result to ~2^-21 (the plain fp16
    output: 2^-12); GroupNorm reads a split pair or the fp32 tensor alike."""
    L = lib()
    g = torch.Generator().manual_seed(11)
    R, C = 300, 640
    x = torch.randn(R, C, generator=g) * 3 + 0.5
    gam, bet = 1 + 0.1 * torch.randn(C, generator=g), 0.1 * torch.randn(C, generator=g)
    ref = F.layer_norm(x.double(), (C,), gam.double(), bet.double(), 1e-5)
    xd, gd, bd = x.cuda(), gam.cuda(), bet.cuda()
    y = torch.zeros(R, 2 * C, dtype=torch.half, device="cuda")
    ok(L.gdf_op_layernorm_split(P(xd), C, R, C, 1e-5, P(gd), P(bd), P(y), 2 * C, C, stream()), L)
    torch.cuda.synchronize()
    pair = y[:, :C].double().cpu() + y[:, C:].double().cpu()
    assert rel(pair, ref) < 3e-6 and rel(y[:, :C], ref) > 1e-4
    # GroupNorm: fp32 input and split-pair input, split output
    B, HW, Cg = 2, 24 * 24, 320
    xg = torch.randn(B, HW, Cg, generator=g) * 2 + 0.3
    gam2, bet2 = 1 + 0.1 * torch.randn(Cg, generator=g), 0.1 * torch.randn(Cg, generator=g)
    refg = F.silu(F.group_norm(xg.permute(0, 2, 1).double(), 32, gam2.double(), bet2.double(), 1e-5)).permute(0, 2, 1)
    hi, lo = _split(xg)
    xp = torch.cat([hi, lo], -1).cuda()                                          # [hi | lo], ld = 2C
    x32d, g2, b2 = xg.cuda(), gam2.cuda(), bet2.cuda()
    scratch = torch.empty(L.gdf_op_groupnorm_scratch_bytes(B, HW, Cg) + 1024, dtype=torch.uint8, device="cuda")
    for src16, x_lo, src32, ld in ((None, 0, x32d, Cg), (xp, Cg, None, 2 * Cg)):
        yg = torch.zeros(B, HW, 2 * Cg, dtype=torch.half, device="cuda")
        ok(L.gdf_op_groupnorm_split(P(src16), x_lo, P(src32), ld, B, HW, Cg, 32, 1e-5, P(g2), P(b2), 1, P(yg), 2 * Cg, Cg, P(scratch), stream()), L)
        torch.cuda.synchronize()
        pg = yg[..., :Cg].double().cpu() + yg[..., Cg:].double().cpu()
        assert rel(pg, refg) < 5e-6, rel(pg, refg)
    # attention: split output
    Bq, heads, S, D = 2, 5, 512, 64
    Ca = heads * D
    qkv = torch.randn(Bq * S, 3 * Ca, generator=g).half()
    q, k, v = [t.float().view(Bq, S, heads, D).transpose(1, 2) for t in qkv.split(Ca, -1)]
    refa = F.scaled_dot_product_attention(q.double(), k.double(), v.double()).transpose(1, 2).reshape(Bq * S, Ca)
    qd = qkv.cuda()
    o = torch.zeros(Bq * S, 2 * Ca, dtype=torch.half, device="cuda")
    ok(L.gdf_op_attention_split(P(qd), 3 * Ca, C_off(qd, Ca), 3 * Ca, C_off(qd, 2 * Ca), 3 * Ca, P(o), 2 * Ca, Ca, Bq, heads, S, S, D, None, stream()), L)
    torch.cuda.synchronize()
    o1 = torch.zeros(Bq * S, Ca, dtype=torch.half, device="cuda")
    ok(L.gdf_op_attention(P(qd), 3 * Ca, C_off(qd, Ca), 3 * Ca, C_off(qd, 2 * Ca), 3 * Ca, P(o1), Ca, Bq, heads, S, S, D, None, stream()), L)
    torch.cuda.synchronize()
    assert torch.equal(o[:, :Ca], o1)                                            # the hi half IS the plain output
    pa = o[:, :Ca].double().cpu() + o[:, Ca:].double().cpu()
    assert rel(pa, refa) < rel(o1, refa)                                         # the pair removes the output rounding (P stays fp16)


@pytest.mark.gpu
@pytest.mark.parametrize("D,S,gain", [(64, 512, 3.0), (64, 320, 1.0), (40, 448, 3.0), (80, 192, 3.0), (160, 64, 1.0)])
def test_attention_over_split_qkv_pairs(D, S, gain):
    """gdf_op_attention_pair (AttnParams::qkv_lo, attn_kernel<..., QKP>): q, k, v as (hi, lo) pairs in the row layout a GEMM with o16_lo writes,
    [q | k | v | q_lo | k_lo | v_lo].  Against fp64 SDPA of the UNROUNDED q, k, v: the pair form removes the storage rounding in front of the
    softmax (what remains is P in fp16 and the output rounding), the plain kernel on the hi halves does not — most visibly with peaked softmaxes
    (`gain` scales q: larger logits).  D = 160 has no pair kernel (its four tiles do not fit in LDS): the call reads the hi halves."""
    L = lib()
    g = torch.Generator().manual_seed(D + S)
    Bq, heads = 2, 3
    Ca = heads * D
    x = torch.randn(Bq * S, 3 * Ca, generator=g)
    x[:, :Ca] *= gain
    hi, lo = _split(x)
    q, k, v = [t.double().view(Bq, S, heads, D).transpose(1, 2) for t in x.split(Ca, -1)]
    ref = F.scaled_dot_product_attention(q, k, v).transpose(1, 2).reshape(Bq * S, Ca)
    pair = torch.cat([hi, lo], -1).cuda()                                        # ld = 6 Ca, lo at + 3 Ca
    o_pair = torch.zeros(Bq * S, 2 * Ca, dtype=torch.half, device="cuda")
    ok(L.gdf_op_attention_pair(P(pair), 6 * Ca, C_off(pair, Ca), 6 * Ca, C_off(pair, 2 * Ca), 6 * Ca, 3 * Ca, P(o_pair), 2 * Ca, Ca, Bq, heads, S, S, D,
                               stream()), L)
    o_plain = torch.zeros(Bq * S, 2 * Ca, dtype=torch.half, device="cuda")
    ok(L.gdf_op_attention_split(P(pair), 6 * Ca, C_off(pair, Ca), 6 * Ca, C_off(pair, 2 * Ca), 6 * Ca, P(o_plain), 2 * Ca, Ca, Bq, heads, S, S, D, None,
                                stream()), L)
    torch.cuda.synchronize()
    e_pair = rel(o_pair[:, :Ca].double().cpu() + o_pair[:, Ca:].double().cpu(), ref)
    e_plain = rel(o_plain[:, :Ca].double().cpu() + o_plain[:, Ca:].double().cpu(), ref)
    print(f"[attention pair] D={D} S={S} gain={gain}: plain (hi halves) {e_plain:.2e}  pair {e_pair:.2e}")
    assert torch.isfinite(o_pair.float()).all()
    if D == 160:
        assert torch.equal(o_pair, o_plain)
    else:
        assert e_pair < 0.6 * e_plain and e_pair < 2.5e-4


def test_gemm_beside_another_kernel_on_a_second_stream_keeps_its_bits():
    """Round 6: the 256x256 two-group main loop restaged B rows in the phase in which two of its waves still read them; harmless with real tiles
    (a DMA needs ~1 us), but the tiles staged past the end of K are out-of-range loads that land in ~100 cycles, and with ANOTHER kernel's waves
    on the CU (a second stream: one extractor per host thread) the reads lost about once in 10^3 launches — 64 rows x 16 columns of a tile
    computed without its last K-tiles (profiles/r06_concurrent_streams.txt).  The GEGLU GEMM of an SD1.5 / SDXL level-0 block (K = 320: 5
    K-tiles) beside a LayerNorm kernel, 4000 launches per thread, every result compared bit for bit.  (The round-5 kernel fails this about
    4 times per run.)"""
    import threading
    L = lib()
    M, C = 4096, 320
    x, W, b = rnd(M, C), rnd(8 * C, C, scale=C ** -0.5), rnd(8 * C).float()
    Wd = torch.empty_like(W, device="cuda"); bd = torch.empty(8 * C, device="cuda")
    Ws, bs, xd = W.cuda(), b.cuda(), x.cuda()
    ok(L.gdf_op_relayout_geglu(P(Ws), P(bs), P(Wd), P(bd), 8 * C, C, 16, stream()), L)
    out = torch.zeros(M, 4 * C, dtype=torch.half, device="cuda")
    ln_x = rnd(M, C, seed=7).float().cuda(); ln_g = rnd(C, seed=8).float().cuda(); ln_b = rnd(C, seed=9).float().cuda()
    ln_y = torch.zeros(M, C, dtype=torch.half, device="cuda")
    torch.cuda.synchronize()
    s0, s1 = torch.cuda.Stream(), torch.cuda.Stream()
    import ctypes
    sp = lambda s: ctypes.c_void_p(s.cuda_stream)

    def gemm():
        ok(L.gdf_op_gemm(P(xd), C, P(Wd), P(bd), None, None, 0, P(out), 4 * C, None, 0, M, 8 * C, C, 1, sp(s0)), L)

    def ln():
        ok(L.gdf_op_layernorm(None, P(ln_x), C, M, C, 1e-5, P(ln_g), P(ln_b), P(ln_y), sp(s1)), L)

    gemm(); s0.synchronize(); base = out.clone()
    ln(); s1.synchronize(); base_ln = ln_y.clone()
    bad = [0, 0]
    bar = threading.Barrier(2)

    def work(i):
        torch.cuda.set_device(0)
        bar.wait()
        for _ in range(4000):
            if i == 0:
                gemm(); s0.synchronize(); bad[0] += int(not torch.equal(out, base))
            else:
                ln(); s1.synchronize(); bad[1] += int(not torch.equal(ln_y, base_ln))
    th = [threading.Thread(target=work, args=(i,)) for i in range(2)]
    [t.start() for t in th]; [t.join() for t in th]
    assert bad == [0, 0], bad
