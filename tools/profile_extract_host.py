"""Host-side profile of FeatureExtractor.extract(image_type='tensors') on SDXL 1024^2 B=16 (cProfile, 10 calls)."""
import cProfile, os, pstats, sys, io
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "generic-diffusion-feature_amd")):
    sys.path.insert(0, p)
os.environ.setdefault("GDF_SYNTHETIC_WEIGHTS", "1")
import torch, time
import diffusion_feature
ids = ["up-level0-repeat0-vit-block7-out", "up-level0-repeat0-vit-block5-out", "up-level1-repeat0-vit-block0-cross-q", "up-level1-repeat0-vit-block0-out"]
df = diffusion_feature.FeatureExtractor({k: True for k in ids}, "xl", device="cuda:0", img_size=1024)
prompts = df.encode_prompt("a photo of a cat")
imgs = (torch.rand(16, 3, 1024, 1024, device="cuda") * 2 - 1).half()
for _ in range(4):
    f = df.extract(prompts, 16, imgs, image_type="tensors", t=100); del f
torch.cuda.synchronize()
pr = cProfile.Profile()
t0 = time.perf_counter(); c0 = time.process_time()
pr.enable()
for _ in range(10):
    f = df.extract(prompts, 16, imgs, image_type="tensors", t=100); del f
pr.disable()
t_host = time.perf_counter() - t0
torch.cuda.synchronize()
print(f"host wall time inside extract(): {1e3 * t_host / 10:.2f} ms per call (GPU work is asynchronous); total with sync {1e3 * (time.perf_counter() - t0) / 10:.1f} ms")
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(22); print(s.getvalue()[:4500])
