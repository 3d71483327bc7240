#!/usr/bin/env python3
"""BASELINE config C1: SD1.5 512x512, 1 image, t=100, up/down-block-out hooks through the CPU path (plumbing, no GPU).

Runs the fp32 CPU oracle (oracle/unet_ref.py — test infrastructure; this tool is the cpu_baseline / plumbing leg, never the
product path) at the TRUE SD1.5 architecture with seeded synthetic weights and the reference's legacy layer selection
(feature/configs/config_15_legacy.json) + the down-block outputs, and prints shapes, order and timing.
    python tools/run_config_c1.py [--lat 64] [--threads 8]"""
import argparse, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from oracle import unet_ref as R

ap = argparse.ArgumentParser()
ap.add_argument("--lat", type=int, default=64); ap.add_argument("--threads", type=int, default=os.cpu_count() or 1)
a = ap.parse_args()
torch.set_num_threads(a.threads)
arch = R.ARCHS["1-5"]
legacy = list(json.load(open(os.path.join(ROOT, "generic-diffusion-feature_amd", "configs", "config_15_legacy.json"))).keys())
ids = legacy + [f"down-level{l}-downsampler-out" for l in range(3)] + ["down-level3-repeat1-res-out"]
t0 = time.time(); P = R.synth_params(arch, seed=0); t_w = time.time() - t0
I = R.synth_inputs(arch, 1, a.lat, seed=1)
st = R.Store({k: True for k in ids})
t0 = time.time()
with torch.no_grad():
    y = R.unet_forward(P, arch, I["sample"], I["timestep"], I["ctx"], store=st)
dt = time.time() - t0
print(json.dumps(dict(config="C1 SD1.5 %dx%d, 1 image, t=100, CPU oracle fp32" % (a.lat * 8, a.lat * 8), threads=a.threads,
                      weights_s=round(t_w, 1), forward_s=round(dt, 2), images_per_s=round(1 / dt, 4),
                      hooks={k: list(v.shape) for k, v in st.feats.items()}, noise_pred=list(y.shape))))
