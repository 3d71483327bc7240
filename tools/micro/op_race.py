#!/usr/bin/env python3
"""Which KERNEL gives different bits when two host threads launch it at the same time on two streams (each thread its own buffers)?
Per op: baseline while the other thread idles, then RACE_ITERS concurrent launches per thread, every result compared bit for bit."""
import os
import sys
import threading

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402
from ops_binding import P, lib, ok  # noqa: E402
import ctypes as C  # noqa: E402

L = lib()
N_ITER = int(os.environ.get("RACE_ITERS", "200"))
torch.cuda.set_device(0)


def rnd(*shape, scale=1.0, seed=0):
    g = torch.Generator().manual_seed(seed + sum(shape))
    return (torch.randn(*shape, generator=g) * scale).half().cuda()


def make_geglu(M, Cc, variant, seed):
    x, W, b = rnd(M, Cc, seed=seed), rnd(8 * Cc, Cc, scale=Cc ** -0.5, seed=seed + 1), rnd(8 * Cc, seed=seed + 2).float()
    Wd = torch.empty_like(W); bd = torch.empty(8 * Cc, device="cuda")
    ok(L.gdf_op_relayout_geglu(P(W), P(b), P(Wd), P(bd), 8 * Cc, Cc, 16, C.c_void_p(0)), L)
    torch.cuda.synchronize()
    out = torch.zeros(M, 4 * Cc, dtype=torch.half, device="cuda")

    def run(s):
        out.fill_(0) if False else None
        ok(L.gdf_op_gemm(P(x), Cc, P(Wd), P(bd), None, None, 0, P(out), 4 * Cc, None, 0, M, 8 * Cc, Cc, 1 | (variant << 8), C.c_void_p(s.cuda_stream)), L)
        return out
    return run


def make_gemm(M, N, K, variant, seed, res32=True):
    A, W = rnd(M, K, seed=seed), rnd(N, K, scale=K ** -0.5, seed=seed + 1)
    bias, res = rnd(N, seed=seed + 2).float(), rnd(M, N, seed=seed + 3).float()
    o16 = torch.zeros(M, N, dtype=torch.half, device="cuda"); o32 = torch.zeros(M, N, device="cuda")

    def run(s):
        ok(L.gdf_op_gemm(P(A), K, P(W), P(bias), P(res) if res32 else None, None, N, P(o16), N, P(o32) if res32 else None, N, M, N, K, variant << 8,
                         C.c_void_p(s.cuda_stream)), L)
        return o16
    return run


def make_ln(R, Cc, seed):
    x = rnd(R, Cc, seed=seed).float()
    g, b = rnd(Cc, seed=seed + 1).float(), rnd(Cc, seed=seed + 2).float()
    y = torch.zeros(R, Cc, dtype=torch.half, device="cuda")

    def run(s):
        ok(L.gdf_op_layernorm(None, P(x), Cc, R, Cc, 1e-5, P(g), P(b), P(y), C.c_void_p(s.cuda_stream)), L)
        return y
    return run


def make_attn(B, heads, S, D, seed):
    q, k, v = rnd(B * S, heads * D, seed=seed), rnd(B * S, heads * D, seed=seed + 1), rnd(B * S, heads * D, seed=seed + 2)
    o = torch.zeros(B * S, heads * D, dtype=torch.half, device="cuda")

    def run(s):
        ok(L.gdf_op_attention(P(q), heads * D, P(k), heads * D, P(v), heads * D, P(o), heads * D, B, heads, S, S, D, None, C.c_void_p(s.cuda_stream)), L)
        return o
    return run


CASES = {
    "geglu M4096 C320 (auto tile)": lambda sd: make_geglu(4096, 320, 0, sd),
    "geglu M4096 C320 tile 825 (256x256 8-phase)": lambda sd: make_geglu(4096, 320, 825, sd),
    "geglu M4096 C320 tile 128": lambda sd: make_geglu(4096, 320, 128, sd),
    "geglu M16384 C1280 (auto)": lambda sd: make_geglu(16384, 1280, 0, sd),
    "gemm res32 4096x320x320 (auto)": lambda sd: make_gemm(4096, 320, 320, 0, sd),
    "gemm res32 4096x320x1280 (auto)": lambda sd: make_gemm(4096, 320, 1280, 0, sd),
    "gemm qkv 4096x960x320 (auto)": lambda sd: make_gemm(4096, 960, 320, 0, sd, res32=False),
    "gemm 16384x1280x1280 tile 932": lambda sd: make_gemm(16384, 1280, 1280, 932, sd, res32=False),
    "layernorm 4096x320": lambda sd: make_ln(4096, 320, sd),
    "attention B4 h8 S1024 D40": lambda sd: make_attn(4, 8, 1024, 40, sd),
}
def make_conv(B, H, W, Cin, Cout, seed):
    x = rnd(B, H, W, Cin, seed=seed)
    w = rnd(Cout, Cin, 3, 3, scale=(9 * Cin) ** -0.5, seed=seed + 1)
    wd = torch.empty(Cout, 9 * Cin, dtype=torch.half, device="cuda")
    ok(L.gdf_op_relayout_conv3(P(w), P(wd), Cout, Cin, C.c_void_p(0)), L)
    torch.cuda.synchronize()
    bias = rnd(Cout, seed=seed + 2).float()
    o16 = torch.zeros(B, H, W, Cout, dtype=torch.half, device="cuda")

    def run(s):
        ok(L.gdf_op_conv3x3(P(x), Cin, B, H, W, Cin, P(wd), Cout, P(bias), None, 1, 0, None, None, P(o16), None, 0, C.c_void_p(s.cuda_stream)), L)
        return o16
    return run


def make_gn(B, HW, Cc, seed):
    x = rnd(B * HW, Cc, seed=seed)
    g, b = rnd(Cc, seed=seed + 1).float(), rnd(Cc, seed=seed + 2).float()
    y = torch.zeros(B * HW, Cc, dtype=torch.half, device="cuda")
    scratch = torch.zeros(int(L.gdf_op_groupnorm_scratch_bytes(B, HW, Cc)) + 16, dtype=torch.uint8, device="cuda")

    def run(s):
        ok(L.gdf_op_groupnorm(P(x), None, Cc, B, HW, Cc, 32, 1e-5, P(g), P(b), 1, P(y), P(scratch), C.c_void_p(s.cuda_stream)), L)
        return y
    return run


CASES["conv3x3 B4 32x32 320->320"] = lambda sd: make_conv(4, 32, 32, 320, 320, sd)
CASES["groupnorm+silu B4 1024x320"] = lambda sd: make_gn(4, 1024, 320, sd)
CASES["gemm 4096x960x320 tile 932 (256x320 8-phase)"] = lambda sd: make_gemm(4096, 960, 320, 932, sd, res32=False)
CASES["gemm res32 4096x640x320 tile 932"] = lambda sd: make_gemm(4096, 640, 320, 932, sd)
CASES["conv3x3 B4 32x32 320->640 (auto: 932)"] = lambda sd: make_conv(4, 32, 32, 320, 640, sd)
def make_dit(M, N, K, variant, seed, act=1):
    A, W = rnd(M, K, seed=seed), rnd(N, K, scale=K ** -0.5, seed=seed + 1)
    bias = rnd(N, seed=seed + 2).float()
    o16 = torch.zeros(M, N, dtype=torch.half, device="cuda")

    def run(s):
        ok(L.gdf_op_gemm_dit(P(A), K, P(W), P(bias), act, None, 0, 0, 1, 0, 1, None, 0, None, 0, P(o16), N, None, 0, M, N, K, variant, C.c_void_p(s.cuda_stream)), L)
        return o16
    return run


def make_joint(B, heads, T, S, D, seed):
    n = T + S
    q, k, v = rnd(B * n, heads * D, seed=seed), rnd(B * n, heads * D, seed=seed + 1), rnd(B * n, heads * D, seed=seed + 2)
    o = torch.zeros(B * n, heads * D, dtype=torch.half, device="cuda")

    def run(s):
        ok(L.gdf_op_attention_joint(P(q), heads * D, P(k), heads * D, P(v), heads * D, P(o), heads * D, B, heads, T, S, D, C.c_void_p(s.cuda_stream)), L)
        return o
    return run


def make_cross(B, heads, Sq, Sk, D, seed):
    q = rnd(B * Sq, heads * D, seed=seed)
    k, v = rnd(B * Sk, heads * D, seed=seed + 1), rnd(B * Sk, heads * D, seed=seed + 2)
    o = torch.zeros(B * Sq, heads * D, dtype=torch.half, device="cuda")

    def run(s):
        ok(L.gdf_op_attention(P(q), heads * D, P(k), heads * D, P(v), heads * D, P(o), heads * D, B, heads, Sq, Sk, D, None, C.c_void_p(s.cuda_stream)), L)
        return o
    return run


CASES["attention cross B4 h8 Sq1024 Sk77 D40"] = lambda sd: make_cross(4, 8, 1024, 77, 40, sd)
CASES["attention self B2 h10 S4096 D64"] = lambda sd: make_cross(2, 10, 4096, 4096, 64, sd)
CASES["attention joint B1 h24 512+1024 D128"] = lambda sd: make_joint(1, 24, 512, 1024, 128, sd)
CASES["attention joint B2 h24 128+1024 D128"] = lambda sd: make_joint(2, 24, 128, 1024, 128, sd)


def make_joint_qkv(B, heads, T, S, D, seed):
    """q | k | v as column blocks of ONE (rows, 3C) buffer: how the MMDiT plan calls the kernel (csrc/flux.cpp joint_attention)"""
    n, Cc = T + S, heads * D
    qkv = rnd(B * n, 3 * Cc, seed=seed)
    o = torch.zeros(B * n, Cc, dtype=torch.half, device="cuda")
    off = lambda k: C.c_void_p(qkv.data_ptr() + k * Cc * 2)

    def run(s):
        ok(L.gdf_op_attention_joint(off(0), 3 * Cc, off(1), 3 * Cc, off(2), 3 * Cc, P(o), Cc, B, heads, T, S, D, C.c_void_p(s.cuda_stream)), L)
        return o
    return run


CASES["attention joint qkv-interleaved B2 h24 128+1024 D128"] = lambda sd: make_joint_qkv(2, 24, 128, 1024, 128, sd)


def make_qknr(R, heads, seed):
    """in-place RMSNorm + RoPE on the q | k column blocks of a (R, 3C) buffer (the text stream of the MMDiT: csrc/dit.hip); the input is restored
    from a pristine copy on the same stream before every launch"""
    D = 128; Cc = heads * D
    src = rnd(R, 3 * Cc, seed=seed)
    x = src.clone()
    wq, wk = (1 + 0.1 * rnd(D, seed=seed + 1).float()).contiguous(), (1 + 0.1 * rnd(D, seed=seed + 2).float()).contiguous()
    ang = torch.rand(R, D, generator=torch.Generator().manual_seed(seed)).cuda() * 6.28
    cos, sin = torch.cos(ang).contiguous(), torch.sin(ang).contiguous()

    def run(s):
        with torch.cuda.stream(s):
            if os.environ.get("RACE_RESTORE", "memcpy") == "kernel":
                torch.add(src, 0, out=x)          # restore by a KERNEL instead of a D2D memcpy
            else:
                x.copy_(src)
        ok(L.gdf_op_qk_norm_rope(P(x), 3 * Cc, R, heads, 0, Cc, P(wq), P(wk), 1e-6, P(cos), P(sin), 0, R, C.c_void_p(s.cuda_stream)), L)
        return x
    return run


def make_lnmod(R, Cc, seed):
    x = rnd(R, Cc, seed=seed).float()
    mod = rnd(2, 2 * Cc, seed=seed + 1).float()
    y = torch.zeros(R, Cc, dtype=torch.half, device="cuda")

    def run(s):
        ok(L.gdf_op_layernorm_mod(P(x), Cc, R, Cc, 1e-6, C.c_void_p(mod.data_ptr() + Cc * 4), P(mod), 2 * Cc, R // 2, R, R // 2, P(y), C.c_void_p(s.cuda_stream)), L)
        return y
    return run


CASES["dit gemm 256x9216x3072 (text qkv, auto tile)"] = lambda sd: make_dit(256, 9216, 3072, 0, sd, act=0)
CASES["dit gemm 256x3072x3072 (text out, auto tile)"] = lambda sd: make_dit(256, 3072, 3072, 0, sd, act=0)
CASES["dit gemm 2304x9216x3072 (single-block qkv, auto)"] = lambda sd: make_dit(2304, 9216, 3072, 0, sd, act=0)
CASES["qk_norm_rope R256 h24 (in place)"] = lambda sd: make_qknr(256, 24, sd)
def make_torch_victim(kind, seed):
    """a kernel that is NOT ours as the victim: torch elementwise / reduction kernels on a (256, 9216) fp16 tensor"""
    a = rnd(256, 9216, seed=seed); b = rnd(256, 9216, seed=seed + 1)
    y = torch.empty_like(a)

    def run(s):
        with torch.cuda.stream(s):
            if kind == "fma":
                torch.addcmul(a, a, b, value=0.5, out=y)
            elif kind == "cos":
                torch.cos(a.float()).half()
                torch.mul(torch.cos(a.float()), b.float()).half()
                y.copy_(torch.mul(torch.cos(a.float()), torch.rsqrt(b.float().abs() + 1.0)))
            else:
                y.copy_(torch.softmax(a.float(), dim=-1))
        return y
    return run


def make_idle(seed):
    """no kernel at all: a (256, 9216) fp16 buffer plus the small tables of the rope case, only READ by the comparison — any change is somebody else's write"""
    x = rnd(256, 9216, seed=seed)
    small = [rnd(256, 128, seed=seed + k).float() for k in range(4)]

    def run(s):
        return x
    return run


def make_torch_matmul(M, N, K, seed):
    A, W = rnd(M, K, seed=seed), rnd(N, K, scale=K ** -0.5, seed=seed + 1)
    o = torch.empty(M, N, dtype=torch.half, device="cuda")

    def run(s):
        with torch.cuda.stream(s):
            torch.matmul(A, W.t(), out=o)
        return o
    return run


CASES["T1 torch matmul 256x9216x3072 (hipBLASLt / rocBLAS)"] = lambda sd: make_torch_matmul(256, 9216, 3072, sd)
CASES["T2 torch matmul 4096x9216x320"] = lambda sd: make_torch_matmul(4096, 9216, 320, sd)
CASES["V0 idle buffer (nobody writes it)"] = lambda sd: make_idle(sd)
CASES["V1 torch addcmul (256x9216 fp16)"] = lambda sd: make_torch_victim("fma", sd)
CASES["V2 torch cos*rsqrt (256x9216)"] = lambda sd: make_torch_victim("cos", sd)
CASES["V3 torch softmax rows (256x9216)"] = lambda sd: make_torch_victim("softmax", sd)
CASES["P1 dit gemm 256x9216x3072 tile 128"] = lambda sd: make_dit(256, 9216, 3072, 128, sd, act=0)
CASES["P2 dit gemm 256x9216x3072 tile 2128"] = lambda sd: make_dit(256, 9216, 3072, 2128, sd, act=0)
CASES["P3 dit gemm 256x9216x3072 tile 8256"] = lambda sd: make_dit(256, 9216, 3072, 8256, sd, act=0)
CASES["P4 unet gemm 256x9216x3072 tile 128"] = lambda sd: make_gemm(256, 9216, 3072, 128, sd, res32=False)
CASES["P5 unet gemm 256x9216x3072 tile 160"] = lambda sd: make_gemm(256, 9280, 3072, 160, sd, res32=False)
CASES["P6 unet gemm 4096x9216x320 tile 128"] = lambda sd: make_gemm(4096, 9216, 320, 128, sd, res32=False)
CASES["layernorm_mod 2304x3072"] = lambda sd: make_lnmod(2304, 3072, sd)
CASES["dit gemm 4096x3072x3072 (8256) gelu"] = lambda sd: make_dit(4096, 3072, 3072, 8256, sd)
CASES["dit gemm 2048x1024x256 (8256) short K"] = lambda sd: make_dit(2048, 1024, 256, 8256, sd)
CASES["conv3x3 B2 64x64 256->256 (826)"] = lambda sd: make_conv(2, 64, 64, 256, 256, sd)
CASES["conv3x3 B1 128x128 128->128"] = lambda sd: make_conv(1, 128, 128, 128, 128, sd)
CASES["geglu M2048 C640 (K=640)"] = lambda sd: make_geglu(2048, 640, 0, sd)
CASES["attention cross placeholder"] = None
only = os.environ.get("RACE_ONLY", "")
pair = os.environ.get("RACE_PAIR", "")          # e.g. "geglu M4096 C320 (auto tile)": thread 0 runs THIS case, thread 1 runs each other case in turn
for name, mk in CASES.items():
    if mk is None or (only and not any(o in name for o in only.split("|"))):
        continue
    runs = [CASES[pair](100) if pair else mk(100), mk(200)]
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    base = []
    for i in range(2):
        for _ in range(2):
            o = runs[i](streams[i]); streams[i].synchronize()
        base.append(o.clone())
    bad = [0, 0]
    detail = [None, None]
    bar = threading.Barrier(2)

    def work(i):
        torch.cuda.set_device(0)
        bar.wait()
        for it in range(N_ITER):
            o = runs[i](streams[i])
            streams[i].synchronize()
            if not torch.equal(o, base[i]):
                bad[i] += 1
                if detail[i] is None:
                    d = (o != base[i])
                    rows = d.any(1).nonzero().flatten()
                    cols = d.any(0).nonzero().flatten()
                    detail[i] = f"iter {it}: {int(d.sum())} elements, rows {int(rows.min())}..{int(rows.max())} ({rows.numel()}), cols {int(cols.min())}..{int(cols.max())} ({cols.numel()})"
    ths = [threading.Thread(target=work, args=(i,)) for i in range(2)]
    [t.start() for t in ths]
    [t.join() for t in ths]
    print(f"{name:<48s} differing launches: {bad[0]:4d} / {bad[1]:4d} of {N_ITER}   {detail[0] or ''} | {detail[1] or ''}", flush=True)
