#!/usr/bin/env python3
"""The CLI's whole loop — JPEG files on disk -> decode / resize / normalise -> VAE encode + UNet + hooks -> D2H -> .npy files — with the reference's
serial input side (load the batch, then extract) and with the loader threads of round 5 (extract_feature.BatchLoader), one FeatureExtractor, SDXL 1024^2,
batch 16, synthetic weights.   python tools/bench_cli.py [--images 96] [--threads 16]"""
import argparse, glob, json, os, shutil, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "generic-diffusion-feature_amd")):
    sys.path.insert(0, p)
import numpy as np
import torch
os.environ.setdefault("GDF_SYNTHETIC_WEIGHTS", "1")
import diffusion_feature
import extract_feature as cli
from PIL import Image

ap = argparse.ArgumentParser(); ap.add_argument("--images", type=int, default=96); ap.add_argument("--threads", type=int, default=min(16, os.cpu_count() or 1))
ap.add_argument("--batch", type=int, default=16); ap.add_argument("--src", type=int, default=1280, help="side of the source JPEGs (resized to --img)")
ap.add_argument("--version", default="xl"); ap.add_argument("--img", type=int, default=0)
ap.add_argument("--no-write", action="store_true", help="diagnostics: drop the features instead of handing them to HostWriter")
a = ap.parse_args()
tmp = tempfile.mkdtemp(prefix="gdf_cli_", dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
try:
    rs = np.random.RandomState(0)
    base = (rs.rand(a.src // 8, a.src // 8, 3) * 255).astype(np.uint8)
    for i in range(a.images):                      # smooth-ish content (upsampled noise + a per-image shift): realistic JPEG decode cost
        im = Image.fromarray(np.roll(base, i, 0)).resize((a.src, int(a.src * 0.75)), Image.BICUBIC)
        im.save(os.path.join(tmp, f"img{i:04d}.jpg"), quality=92)
    paths = sorted(glob.glob(os.path.join(tmp, "*.jpg")))
    import bench as BB
    ids = BB.PRACTICAL[a.version]
    img_size = a.img or (1024 if a.version == "xl" else 512)
    df = diffusion_feature.FeatureExtractor({k: True for k in ids}, a.version, device="cuda:0", img_size=img_size)
    prompts = df.encode_prompt("a photo of a cat")
    out = {"version": a.version, "img_size": img_size, "images": a.images, "batch": a.batch, "source": f"{a.src}x{int(a.src * 0.75)} JPEG q92", "host_cpus": os.cpu_count()}

    def run(threads):
        odir = os.path.join(tmp, f"out{threads}")
        w = cli.HostWriter(argparse.Namespace(output_dir=odir, aggregate_output=False, sample_name_first=False))
        starts = list(range(0, len(paths), a.batch))
        loader = cli.BatchLoader(paths, starts, len(paths), a.batch, df.preprocess_image, threads) if threads > 0 else None
        torch.cuda.synchronize(); t0 = time.perf_counter()
        with torch.no_grad():
            for i in starts:
                chunk = paths[i:i + a.batch]
                if loader is not None:
                    feats = df.extract(prompts, len(chunk), loader.get(i), image_type="tensors", t=100)
                    loader.done(i)
                else:
                    feats = df.extract(prompts, len(chunk), [Image.open(p) for p in chunk], t=100)
                if not a.no_write:
                    w.submit(feats, [f"train{i + j}" for j in range(len(chunk))])
                del feats
        if loader is not None:
            loader.close()
        w.close(); torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        files = sorted(glob.glob(os.path.join(odir, "*", "*.npy")))
        return dt, files

    run(0 if a.images <= 32 else a.threads)          # warm-up: plans, graphs, pinned pools
    os.environ["GDF_PREPROCESS_THREADS"] = "0"               # the reference's loop: one image after the other, then the GPU
    dt0, f0 = run(0)
    os.environ.pop("GDF_PREPROCESS_THREADS")
    dt1, f1 = run(a.threads)
    same = len(f0) == len(f1) and all(np.array_equal(np.load(x).view(np.uint16), np.load(y).view(np.uint16)) for x, y in zip(f0[::7], f1[::7]))
    out.update({"serial_input_images_per_s": round(a.images / dt0, 2), "loader_threads": a.threads, "loader_threads_images_per_s": round(a.images / dt1, 2),
                "files_bit_identical": bool(same), "npy_files": len(f1)})
    print(json.dumps(out))
finally:
    shutil.rmtree(tmp, ignore_errors=True)
