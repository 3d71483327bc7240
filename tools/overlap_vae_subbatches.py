#!/usr/bin/env python3
"""VAE encode, 16 x 1024^2: one plan walking its four sub-batches in order, against TWO encoders (8 images each) on two streams whose op programs run
side by side — the conv grids are many rounds deep (16384 tiles on 512 slots), so the HBM-bound ops of one chain (GroupNorm apply, epilogue-heavy
convs) could fill in beside the MFMA-bound main loops of the other.  img/s.
    python tools/overlap_vae_subbatches.py [--reps 4] [--offset_ms 0]"""
import argparse, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "generic-diffusion-feature_amd")):
    sys.path.insert(0, p)
import torch
from components.native import NativeVAEEncoder, VAE_CONFIGS

ap = argparse.ArgumentParser(); ap.add_argument("--reps", type=int, default=4)
a = ap.parse_args()
g = torch.Generator(device="cuda").manual_seed(0)
B = 16
x = (torch.rand(B, 3, 1024, 1024, device="cuda", generator=g) * 2 - 1).half()
eps = torch.randn(B, 4, 128, 128, device="cuda", generator=g).half(); noise = torch.randn_like(eps)
def kw(lo, hi):
    return dict(eps=eps[lo:hi], noise=noise[lo:hi], scaling_factor=0.13025, noise_a=1.0, noise_b=0.6, input_scale=0.86)
one = NativeVAEEncoder(VAE_CONFIGS["sd"], device="cuda:0").init_synthetic(0)
two = [NativeVAEEncoder(VAE_CONFIGS["sd"], device="cuda:0").init_synthetic(0) for _ in range(2)]
streams = [torch.cuda.Stream(), torch.cuda.Stream()]
ref = one.encode(x, **kw(0, B)).clone()
for h in range(2):
    with torch.cuda.stream(streams[h]):
        o = two[h].encode(x[8 * h:8 * h + 8], **kw(8 * h, 8 * h + 8))
        assert torch.equal(o, ref[8 * h:8 * h + 8])
torch.cuda.synchronize()

def serial(n):
    t0 = time.perf_counter()
    for _ in range(n):
        one.encode(x, **kw(0, B))
    torch.cuda.synchronize()
    return B * n / (time.perf_counter() - t0)

def parallel(n):
    t0 = time.perf_counter()
    for _ in range(n):
        for h in range(2):
            with torch.cuda.stream(streams[h]):
                two[h].encode(x[8 * h:8 * h + 8], **kw(8 * h, 8 * h + 8))
    torch.cuda.synchronize()
    return B * n / (time.perf_counter() - t0)

for r in range(3):
    print(f"run {r}: one chain {serial(a.reps):6.1f} img/s   two chains on two streams {parallel(a.reps):6.1f} img/s", flush=True)
