"""Op-level companion of tools/find_batch_divergence.py: which KERNEL gives different bits to identical samples of one batch?
GroupNorm and the 3x3 conv (every tile variant) on one sample repeated B times."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "generic-diffusion-feature_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import torch  # noqa: E402
from ops_binding import P, lib, ok, stream  # noqa: E402

L = lib()
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4
g = torch.Generator().manual_seed(0)


def rep(t):
    return t[:1].expand(B, *t.shape[1:]).contiguous()


def report(name, y):
    nd = [int((y[i] != y[0]).sum()) for i in range(1, B)]
    print(f"{name:60s} differing elements vs sample 0: {nd}")


for (H, C) in ((64, 320), (32, 640), (128, 320)):
    HW = H * H
    x = rep((torch.randn(1, HW, C, generator=g) * 2 + 0.5).half()).cuda()
    gamma, beta = (1 + 0.1 * torch.randn(C, generator=g)).cuda(), (0.1 * torch.randn(C, generator=g)).cuda()
    y = torch.zeros(B, HW, C, dtype=torch.half, device="cuda")
    scratch = torch.zeros(L.gdf_op_groupnorm_scratch_bytes(B, HW, C) + 1024, dtype=torch.uint8, device="cuda")
    ok(L.gdf_op_groupnorm(P(x), None, C, B, HW, C, 32, 1e-5, P(gamma), P(beta), 1, P(y), P(scratch), stream()), L)
    torch.cuda.synchronize()
    report(f"groupnorm+silu {H}x{H} C={C}", y)
    for Cout in (C,):
        w = (torch.randn(Cout, C, 3, 3, generator=g) * (9 * C) ** -0.5).half().cuda()
        wd = torch.empty(Cout, 9 * C, dtype=torch.half, device="cuda")
        ok(L.gdf_op_relayout_conv3(P(w), P(wd), Cout, C, stream()), L)
        bias = torch.randn(Cout, generator=g).cuda()
        temb = rep(torch.randn(1, Cout, generator=g)).cuda()
        xin = y.view(B, H, H, C)
        for variant in (0, 128, 160, 256, 320, 932, 826):
            o16 = torch.zeros(B, H, H, Cout, dtype=torch.half, device="cuda")
            rc = L.gdf_op_conv3x3(P(xin), C, B, H, H, C, P(wd), Cout, P(bias), P(temb), 1, 0, None, None, P(o16), None, variant << 8, stream())
            torch.cuda.synchronize()
            if rc != 0:
                print(f"conv3x3 {H}x{H} {C}->{Cout} variant {variant}: not available ({L.gdf_last_error().decode()[:60]})")
                continue
            report(f"conv3x3 {H}x{H} {C}->{Cout} variant {variant}", o16)

# ---- small linears on identical rows (time_embedding / stacked time_emb_proj) ----
for (K, N, silu) in ((320, 1280, 0), (1280, 1280, 1), (1280, 18560, 1), (1280, 640, 1)):
    xr = rep(torch.randn(1, K, generator=g)).cuda()
    w = (torch.randn(N, K, generator=g) * K ** -0.5).half().cuda()
    bias = torch.randn(N, generator=g).cuda()
    out = torch.zeros(B, N, device="cuda")
    ok(L.gdf_op_small_linear(P(xr), K, B, K, P(w), P(bias), N, silu, 0, P(out), N, stream()), L)
    torch.cuda.synchronize()
    report(f"small_linear K={K} N={N} silu_in={silu}", out)

# ---- the conv2 form: aux16 (res-increment) + fp32 residual in + fp32 master out + fp16 image, strided output (concat slice) ----
H, C = 64, 320
x = rep((torch.randn(1, H, H, C, generator=g)).half()).cuda()
w = (torch.randn(C, C, 3, 3, generator=g) * (9 * C) ** -0.5).half().cuda()
wd = torch.empty(C, 9 * C, dtype=torch.half, device="cuda")
ok(L.gdf_op_relayout_conv3(P(w), P(wd), C, C, stream()), L)
bias = torch.randn(C, generator=g).cuda()
res = rep(torch.randn(1, H, H, C, generator=g)).cuda()
for variant in (0, 320, 932):
    aux = torch.zeros(B, H, H, C, dtype=torch.half, device="cuda"); o16 = torch.zeros_like(aux); o32 = torch.zeros(B, H, H, C, device="cuda")
    rc = L.gdf_op_conv3x3(P(x), C, B, H, H, C, P(wd), C, P(bias), None, 1, 0, P(res), P(aux), P(o16), P(o32), variant << 8, stream())
    torch.cuda.synchronize()
    if rc == 0:
        report(f"conv2 form variant {variant}: aux16", aux); report(f"conv2 form variant {variant}: out16", o16); report(f"conv2 form variant {variant}: out32", o32)
# ---- GroupNorm reading a channel slice of a wider buffer (skip-concat view: ld > C) ----
for ld in (640, 960):
    xw = rep((torch.randn(1, H * H, ld, generator=g) * 2 + 0.5).half()).cuda()
    gamma, beta = (1 + 0.1 * torch.randn(C, generator=g)).cuda(), (0.1 * torch.randn(C, generator=g)).cuda()
    y = torch.zeros(B, H * H, C, dtype=torch.half, device="cuda")
    scratch = torch.zeros(L.gdf_op_groupnorm_scratch_bytes(B, H * H, C) + 1024, dtype=torch.uint8, device="cuda")
    ok(L.gdf_op_groupnorm(P(xw), None, ld, B, H * H, C, 32, 1e-5, P(gamma), P(beta), 1, P(y), P(scratch), stream()), L)
    torch.cuda.synchronize()
    report(f"groupnorm+silu 64x64 C=320 ld={ld}", y)
