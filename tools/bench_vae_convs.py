import sys, torch
sys.path.insert(0, "tests")
from ops_binding import P, lib, ok, stream
L = lib()
def t(fn, it=5):
    for _ in range(2): fn()
    torch.cuda.synchronize(); e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True); e0.record()
    for _ in range(it): fn()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) / it
for name, B, H, Ci, Co in (("128@1024", 4, 1024, 128, 128), ("128->256@512", 4, 512, 128, 256), ("256@512", 4, 512, 256, 256), ("256->512@256", 4, 256, 256, 512), ("512@256", 4, 256, 512, 512), ("512@128", 4, 128, 512, 512)):
    x = torch.randn(B, H, H, Ci, device="cuda").half(); w = (torch.randn(Co, 9 * Ci, device="cuda") * (9 * Ci) ** -0.5).half()
    bias = torch.randn(Co, device="cuda"); o16 = torch.empty(B, H, H, Co, device="cuda", dtype=torch.half)
    out = []
    for var in (128, 160, 256, 320, 826):
        if ((var in (160, 320)) and Co % var) or (var == 826 and Co % 256): out.append("      -      "); continue
        ms = t(lambda: ok(L.gdf_op_conv3x3(P(x), Ci, B, H, H, Ci, P(w), Co, P(bias), None, 1, 0, None, None, P(o16), None, var << 8, stream()), L))
        out.append(f"{ms:6.3f} {2.0 * B * H * H * Co * 9 * Ci / ms / 1e9:6.0f}")
    print(f"{name:14s} 128x128 {out[0]} | 128x160 {out[1]} | 256x128 {out[2]} | 256x320 {out[3]} | 8ph256 {out[4]}")
