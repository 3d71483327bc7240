#!/usr/bin/env python3
"""Soak run of the product call (MI355X box): for `--minutes` a stream of FeatureExtractor lifetimes with random versions / image sizes /
layer subsets / batch sizes / timesteps / feature_resize / 'vae-out' / operand plans, from `--threads` host threads at once (one extractor per
thread at a time: correspondence/correspondence/aggregation_network.py:34-66 keeps several alive side by side), checking what a long-lived
service needs from the library:

  * every extract() repeated on the same inputs returns the same BITS (no state leaks between calls / plans / threads);
  * a fixed probe configuration re-run every few lifetimes returns the bits it returned at the start of the soak;
  * no plan ever fell back from hipGraph replay to eager launches (gdf_plan_graph_failures == 0);
  * device memory comes back: free HBM after gc at the end is within `--leak-mb` of what it was after the first lifetime of each thread;
  * nothing raises, nothing hangs (the caller runs this under `timeout`).

Prints one JSON line.  Synthetic weights (GDF_SYNTHETIC_WEIGHTS=1): no checkpoints exist offline."""
import argparse
import gc
import json
import os
import random
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "generic-diffusion-feature_amd"))
os.environ.setdefault("GDF_SYNTHETIC_WEIGHTS", "1")

import torch  # noqa: E402

# bisection switches (diagnostics): SOAK_LOCK=extract|ctor|all serialises that part of a lifetime across threads; SOAK_SIMPLE=1 keeps to one small
# configuration family (SD1.5, automatic plan, latents in, no 'vae-out')
_LOCKS = {k: threading.Lock() for k in ("extract", "ctor")}
_LOCK_MODE = os.environ.get("SOAK_LOCK", "")
_SIMPLE = os.environ.get("SOAK_SIMPLE", "0") == "1"
_NO_DIT = os.environ.get("SOAK_NO_DIT", "0") == "1"


def _install_fine_locks():
    """SOAK_LOCK=ccall: one lock around the C call of a forward (gdf_forward / gdf_vae_* / ...); SOAK_LOCK=planrun: around _Plan.run (staging + call)"""
    from components import native as N
    lk = threading.Lock()
    if _LOCK_MODE == "planrun":
        orig = N._Plan.run

        def run(self, *a, **k):
            with lk:
                return orig(self, *a, **k)
        N._Plan.run = run
    if _LOCK_MODE.startswith("run_"):
        # a copy of _Plan.run with the lock around a chosen span: run_pre (stream wait + staging), run_precall (+ the C call), run_post (the
        # caller-stream wait + lease), run_callpost
        import ctypes as C

        def run(self, dev, inputs, out_shape, call, profile=False, eager=False, out_dtype=torch.float16):
            spans = {"run_pre": (1, 0, 0), "run_precall": (1, 1, 0), "run_post": (0, 0, 1), "run_callpost": (0, 1, 1), "run_call": (0, 1, 0)}[_LOCK_MODE]
            held = [False]

            def want(i):
                if spans[i] and not held[0]:
                    lk.acquire(); held[0] = True
                if not spans[i] and held[0]:
                    lk.release(); held[0] = False
            try:
                want(0)
                cur = torch.cuda.current_stream(dev)
                if self.stream is None:
                    self.stream = self._make_stream(dev)
                side = self.stream
                side.wait_stream(cur)
                n_out = 1
                for d in out_shape:
                    n_out *= d
                with torch.cuda.device(dev), torch.cuda.stream(side):
                    if self.workspace is None or self.workspace.numel() < self.ws_bytes:
                        self.workspace = torch.empty(self.ws_bytes, dtype=torch.uint8, device=dev)
                    staged = [self._stage(n, t, dt, dev) for (n, t, dt) in inputs]
                    hs = next((h for h in self.sets if h.free()), None)
                    if hs is None:
                        hs = N._HookSet(self, n_out, dev)
                        self.sets.append(hs)
                    want(1)
                    ret = call(staged, hs.ptrs, hs.out_ptr, C.c_void_p(self.workspace.data_ptr()), C.c_void_p(side.cuda_stream))
                want(2)
                cur.wait_stream(side)
                base = hs.lease(dev)
                feats = {}
                for off, (hid, shape, stride, _) in zip(hs.offs, self.hooks):
                    feats[hid] = torch.as_strided(base, shape, stride, storage_offset=off)
                out = base[hs.out_off:hs.out_off + n_out].view(out_dtype).view(out_shape)
                return out, feats, ret
            finally:
                if held[0]:
                    lk.release()
        N._Plan.run = run
    if os.environ.get("SOAK_TCACHE", "") == "off":             # diagnostics: no shared timestep constants
        def fresh(timestep, B, dev):
            if torch.is_tensor(timestep) and timestep.is_cuda:
                t = timestep.to(dev).float().reshape(-1)
                return t.expand(B) if t.numel() == 1 else t
            vals = [float(v) for v in torch.as_tensor(timestep).reshape(-1).tolist()]
            return torch.tensor(vals * (B if len(vals) == 1 else 1), dtype=torch.float32, device=dev)
        N._timestep_on_device = fresh
    if _LOCK_MODE == "ccall":
        orig_launch = N._NativeModel._launch

        def _launch(self, plan, fwd, prof_fn, what, profile):
            def locked(*a):
                with lk:
                    return fwd(*a)
            return orig_launch(self, plan, locked, prof_fn, what, profile)
        N._NativeModel._launch = _launch


class _maybe:
    def __init__(self, what):
        self.l = _LOCKS[what] if _LOCK_MODE in (what, "all") else None

    def __enter__(self):
        if self.l:
            self.l.acquire()

    def __exit__(self, *a):
        if self.l:
            self.l.release()


def all_ids(version):
    from components import feature_extractor as FX
    from components.native import ARCH_CONFIGS, PIXART_CONFIGS
    if version.startswith("pixart"):
        return [i for i in FX.dit_layer_ids(PIXART_CONFIGS[version])]
    return [i for i in FX.unet_layer_ids(ARCH_CONFIGS[version])]


def one_lifetime(rng, dev, stats, probe=None):
    import diffusion_feature
    if probe is None:
        version = rng.choice(["1-5", "1-5", "1-5", "xl", "xl", "2-1", "pixart-sigma-512"] if not _NO_DIT else ["1-5", "1-5", "xl", "2-1"])
        dit = version.startswith("pixart")
        img = rng.choice([256, 512] if dit else [128, 192, 256, 320] if version != "xl" else [256, 384, 512])
        ids = all_ids(version)
        k = rng.randint(1, 12)
        layer = {i: True for i in rng.sample(ids, k)}
        if rng.random() < 0.25 and not dit:
            layer["vae-out"] = True
        kw = dict(feature_resize=rng.choice([1, 1, 2]), early_exit=rng.random() < 0.3)
        if not dit:
            kw["precise"] = rng.choice([None, None, False, True, "selective"])
            if rng.random() < 0.15:
                kw["attention"] = rng.sample(["up_cross", "down_cross", "mid_cross", "up_self", "down_self"], 2)
        calls = rng.randint(1, 4)
        seed = rng.randrange(1 << 30)
    else:
        version, img, layer, kw, calls, seed = probe
    if _SIMPLE and probe is None:
        version, img, kw = "1-5", 256, dict(feature_resize=1)
        layer = {i: True for i in rng.sample(all_ids("1-5"), 4)}
    with _maybe("ctor"):
        df = diffusion_feature.FeatureExtractor(layer=dict(layer), version=version, device=dev, img_size=img, **kw)
        prompt = df.encode_prompt("a photo of a cat")
        torch.cuda.synchronize()
    out = None
    r2 = random.Random(seed)
    for c in range(calls):
        B = r2.randint(1, 5)
        t = r2.choice([1, 50, 100, 261, 500, 999])
        g = torch.Generator().manual_seed(seed + c)
        # 'tensors' goes through the VAE stage, whose eps / noise draws come from the GLOBAL CUDA generator: with several threads another thread's
        # draws interleave, so the bit comparison of that mode is only made single-threaded; 'latents' (no random numbers) is compared always
        via_vae = r2.random() < 0.5 and probe is None and not _SIMPLE
        x = (torch.rand(B, 3, img, img, generator=g) * 2 - 1) if via_vae else torch.randn(B, 4, img // 8, img // 8, generator=g).half()
        res = []
        for rep in range(2):
            with _maybe("extract"):
                torch.manual_seed(seed + c)
                f = df.extract(prompt, batch_size=B, image=x.to(dev), image_type="tensors" if via_vae else "latents", t=t)
                torch.cuda.synchronize()
                res.append({k: v.clone() for k, v in f.items()})
                torch.cuda.synchronize()
        for k in res[0]:
            if (not via_vae or stats["threads"] == 1) and not torch.equal(res[0][k], res[1][k]):
                a_, b_ = res[0][k].float(), res[1][k].float()
                nbad = int((a_ != b_).sum())
                rows = (a_ != b_).reshape(a_.shape[0], -1).any(1).tolist()
                stats["mismatch"].append(f"{version} {img} B={B} t={t} {k}: rel {float((a_ - b_).norm() / (b_.norm() + 1e-30)):.2e}, {nbad}/{a_.numel()} elements, samples {rows}")
            if not torch.isfinite(res[0][k].float()).all():
                stats["nonfinite"].append(f"{version} {img} B={B} t={t} {k}")
        out = res[0]
        stats["extracts"] += 2
        stats["images"] += 2 * B
    u = df.pipe.unet
    for p in list(getattr(u, "_plans", {}).values()):
        cap, lau, fail = p.graph_stats()
        stats["graph_failures"] += fail
        stats["graph_captures"] += cap
    stats["lifetimes"] += 1
    del df, u, prompt
    return out


def worker(idx, args, stats, stop_at, errors):
    try:
        dev = "cuda:0"
        torch.cuda.set_device(0)
        rng = random.Random(args.seed + idx)
        probe = ("1-5", 256, {"down-level1-repeat1-vit-block0-out": True, "mid-vit-block0-ffn-inner": True, "up-level2-repeat1-res-out": True, "unet-out": True},
                 dict(feature_resize=1), 2, 1234 + idx)
        first = one_lifetime(rng, dev, stats, probe)
        n = 0
        while time.time() < stop_at:
            one_lifetime(rng, dev, stats)
            n += 1
            if n % 6 == 0:
                again = one_lifetime(rng, dev, stats, probe)
                stats["probe_runs"] += 1
                if any(not torch.equal(first[k], again[k]) for k in first):
                    stats["probe_drift"].append(f"thread {idx} after {n} lifetimes")
            if n == 1:
                gc.collect()
                stats.setdefault("free_after_first", {})[idx] = torch.cuda.mem_get_info(0)[0]
    except Exception as e:                                    # noqa: BLE001
        import traceback
        errors.append(f"thread {idx}: {type(e).__name__}: {e}\n{traceback.format_exc()[-1500:]}")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--minutes", type=float, default=5.0)
    ap.add_argument("--threads", type=int, default=2)
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--leak-mb", type=float, default=512.0)
    args = ap.parse_args()
    assert torch.cuda.is_available(), "needs an MI355X"
    _install_fine_locks()
    torch.cuda.init()
    free0 = torch.cuda.mem_get_info(0)[0]
    stats = dict(threads=args.threads, extracts=0, images=0, lifetimes=0, probe_runs=0, graph_failures=0, graph_captures=0, mismatch=[], nonfinite=[], probe_drift=[])
    errors = []
    t0 = time.time()
    stop_at = t0 + 60.0 * args.minutes
    ths = [threading.Thread(target=worker, args=(i, args, stats, stop_at, errors)) for i in range(args.threads)]
    for th in ths:
        th.start()
    for th in ths:
        th.join()
    gc.collect()
    torch.cuda.empty_cache()
    torch.cuda.synchronize()
    free1 = torch.cuda.mem_get_info(0)[0]
    leak_mb = (free0 - free1) / 2 ** 20
    ok = not errors and not stats["mismatch"] and not stats["nonfinite"] and not stats["probe_drift"] and stats["graph_failures"] == 0 and leak_mb <= args.leak_mb
    print(json.dumps(dict(ok=ok, minutes=round((time.time() - t0) / 60, 2), threads=args.threads, lifetimes=stats["lifetimes"], extracts=stats["extracts"],
                          images=stats["images"], probe_runs=stats["probe_runs"], graph_captures=stats["graph_captures"], graph_failures=stats["graph_failures"],
                          mismatches=stats["mismatch"][:5], nonfinite=stats["nonfinite"][:5], probe_drift=stats["probe_drift"][:5],
                          free_hbm_mb_start=round(free0 / 2 ** 20), free_hbm_mb_end=round(free1 / 2 ** 20), not_returned_mb=round(leak_mb, 1), errors=errors[:3])))
    sys.exit(0 if ok else 1)


if __name__ == "__main__":
    main()
