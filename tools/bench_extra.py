"""Extra legs of bench.py's JSON line (N = 1, after the headline's timed region; each is bounded to a few seconds):

  plans           the headline step under each operand-plan level (plain = what the headline's four hooks select / selective / precise)
  other_configs   the other BASELINE.json configs under the driver's eyes: SD1.5 512^2 B = 32 (practical hooks and the full 197-id set of
                  config C2, auto-selected plan), Flux.1-dev C5 bf16 B = 8
  e2e             the product call FeatureExtractor.extract(image_type='tensors') on synthetic 1024^2 images: VAE encode + noise-add + UNet +
                  hooks (reference feature/diffusion_feature.py:358-380, :405-465), and the CLI's D2H + .npy stage behind it
                  (reference extract_feature.py:113-148)
  power           package power / shader clock over the timed region (hwmon): the step runs at the 1400-W cap
"""
import glob
import os
import shutil
import tempfile
import threading
import time

import torch


# ---- power / clock sampling (sysfs hwmon of the card behind HIP device `dev`) ------------------------------------------------------
def _bus_id(dev_index):
    try:
        p = torch.cuda.get_device_properties(dev_index)
        return "%04x:%02x:%02x" % (p.pci_domain_id, p.pci_bus_id, p.pci_device_id)
    except Exception:
        return None


def _hwmon(dev_index):
    bus = _bus_id(dev_index)
    for h in glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*"):
        devp = os.path.realpath(os.path.dirname(os.path.dirname(h))).lower()
        if bus and bus.lower() not in devp:
            continue
        if os.path.exists(os.path.join(h, "power1_input")):
            return h
    return None


def _rd(p):
    try:
        with open(p) as f:
            return int(f.read().strip())
    except Exception:
        return None


def _max_sclk_mhz(h):
    """highest shader-clock DPM level of the card behind hwmon dir `h` (pp_dpm_sclk: '0: 132Mhz\n1: 2400Mhz *'), None if unreadable"""
    try:
        with open(os.path.join(os.path.dirname(os.path.dirname(h)), "pp_dpm_sclk")) as f:
            v = [int(l.split(":")[1].strip().lower().split("mhz")[0]) for l in f.read().splitlines() if ":" in l]
        return max(v) if v else None
    except Exception:
        return None


class PowerSampler(threading.Thread):
    """hwmon power / clock sampling.  The sensor averages over ~0.5 s, so a 100-ms cadence loses nothing and wakes the interpreter 10x per
    second instead of 50x (a 20-ms Python thread contends for the GIL with the launching thread inside the timed region)."""

    def __init__(self, dev_index=0, dt=0.1):
        super().__init__(daemon=True)
        self.h = _hwmon(dev_index)
        self.dt = dt
        self.rows = []
        self.stop_flag = False

    def run(self):
        if not self.h:
            return
        pw, fq = os.path.join(self.h, "power1_input"), os.path.join(self.h, "freq1_input")
        while not self.stop_flag:
            self.rows.append((time.perf_counter(), _rd(pw), _rd(fq)))
            time.sleep(self.dt)

    def finish(self, t0, t1, images=None):
        self.stop_flag = True
        if self.is_alive():
            self.join()
        if not self.h:
            return None
        rows = [r for r in self.rows if t0 + 0.25 * (t1 - t0) <= r[0] <= t1 and r[1] is not None]   # the sensor averages over ~0.5 s
        if not rows:
            return None
        cap = _rd(os.path.join(self.h, "power1_cap"))
        w = [r[1] / 1e6 for r in rows]
        f = [r[2] / 1e6 for r in rows if r[2]]
        wavg = sum(w) / len(w)
        return {"watts_avg": round(wavg, 1), "watts_max": round(max(w), 1), "cap_watts": round(cap / 1e6, 1) if cap else None,
                "sclk_mhz_avg": round(sum(f) / len(f)) if f else None, "sclk_mhz_max": _max_sclk_mhz(self.h), "samples": len(rows),
                "joules_per_image": round(wavg * (t1 - t0) / images, 2) if images else None,
                "note": "hwmon power1_input / freq1_input of this card over the last 75 % of the timed region: every MFMA kernel of the step runs at "
                        "the package power cap with the shader clock throttled (profiles/r04_power_per_kernel.txt), so images/s follow ENERGY per "
                        "image, not cycles"}


def _time_steps(step, steps, warm):
    for _ in range(warm):
        o = step(); del o
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        o = step(); del o
    torch.cuda.synchronize()
    return time.perf_counter() - t0


# ---- plan levels ----------------------------------------------------------------------------------------------------------------
def plans_block(unet, step, B, headline_ips, steps=4):
    """the SAME step (same four hooks) under each operand-plan level: what a hook set that selects that level costs"""
    from components.native import SELECTIVE_BY_ARCH, arch_family, SPLIT_SELECTIVE
    out = {"auto_for_these_hooks": {"split_mask": unet.last_split, "images_per_s": round(headline_ips, 2)}}
    sel = SELECTIVE_BY_ARCH.get(arch_family(unet.cfg), SPLIT_SELECTIVE)
    for name, spec in (("selective", sel), ("precise", True)):
        unet.set_precise(spec)
        dt = _time_steps(step, steps, 3)
        out[name] = {"split_mask": unet.last_split, "images_per_s": round(B * steps / dt, 2), "ms_per_step": round(1e3 * dt / steps, 2)}
    unet.set_precise("auto")
    out["note"] = ("operand plan levels (include/gdf.h reserved[1]): auto = the cheapest level that keeps every REQUESTED hook within 1e-3 of the fp32 "
                   "reference (components/plan_levels.py choose_split: plain fp16 operands for the headline's four hooks); selective = split stream "
                   "images (shortcut / proj_out / downsampler operands, GroupNorm inputs) + proj_in / conv_out operands + self-attention outputs (every hook kind <= 8.2e-4 at full size, "
                   "tests/test_gpu_fullsize.py); precise = every operand class split (<= 3e-4 asserted, 1.9e-4 measured: q / k / v pairs since round 5)")
    return out


# ---- other BASELINE configs ---------------------------------------------------------------------------------------------------------
def _unet_inputs(cfg, B, lat, img, dev, seed=7):
    g = torch.Generator(device=dev).manual_seed(seed)
    x = torch.randn(B, 4, lat, lat, generator=g, device=dev).half()
    ctx = torch.randn(1, 77, cfg["cross_attention_dim"], generator=g, device=dev).half().expand(B, -1, -1).contiguous()
    t = torch.full((B,), 100.0, device=dev)
    txt = tid = None
    if cfg["addition_embed_text_time"]:
        pooled = cfg["add_in_dim"] - 6 * cfg["addition_time_embed_dim"]
        txt = torch.randn(1, pooled, generator=g, device=dev).half().expand(B, -1).contiguous()
        tid = torch.tensor([[img, img, 0, 0, img, img]], dtype=torch.float32, device=dev).repeat(B, 1)
    return x, t, ctx, txt, tid


FLUX_EXTRA_DTYPES = ["auto"]      # round 5: the product default (fp16 operands + range-scaled MLP hidden tensors)


def other_configs_block(dev, flops_fn, practical, budget_s=75.0):
    from components.native import NativeUNet, ARCH_CONFIGS, FLUX_CONFIGS, NativeFluxTransformer
    out = {}
    t_start = time.time()
    # ---- BASELINE configs[1]: SD1.5 512^2, batch 32, t = 100: practical hooks and the FULL 197-id layer set (incl. 32 attention maps) ----
    try:
        cfg = ARCH_CONFIGS["1-5"]
        net = NativeUNet(cfg, device=dev).init_synthetic(seed=0)
        B, lat = 32, 64
        ins = _unet_inputs(cfg, B, lat, 512, dev)
        fl_img = sum(flops_fn(cfg, lat).values())
        for tag, ids, steps in (("sd15_512_b32_practical_hooks", practical["1-5"], 5), ("sd15_512_b32_full_layer_set_config_c2", net.hook_names(), 2)):
            step = lambda: net.forward_raw(*ins, hook_ids=ids, shared_ctx=True)
            dt = _time_steps(step, steps, 3)
            o = step(); hb = sum(v.numel() * 2 for v in o[1].values()); del o
            out[tag] = {"images_per_s": round(B * steps / dt, 1), "ms_per_step": round(1e3 * dt / steps, 2), "hooks": len(ids),
                        "hook_gb_per_step": round(hb / 1e9, 2), "split_mask_auto": net.last_split,
                        "model_tflops_per_s": round(B * steps / dt * fl_img / 1e12, 1), "steps": steps}
            net._plans.clear(); torch.cuda.empty_cache()
        del net
        torch.cuda.empty_cache()
    except Exception as e:                                     # an extra leg must never cost the headline line
        out["sd15_error"] = repr(e)[:300]
    # ---- BASELINE configs[4]: Flux.1-dev MMDiT 1024^2, batch 8, every shipped arithmetic mode (tools/bench_flux.py is the full per-model bench):
    # bf16 = the reference's dtype (hooks <= 3.4e-3 at full depth), bfloat16x2 = bf16 hi + lo operand pairs (<= 1.8e-4), fp8-mx = the optional
    # e4m3 MFMA leg (<= 7.5e-2, opt-in, lower precision) ----
    for dt_name in ("bfloat16", "bfloat16x2", "fp8-mx") + tuple(FLUX_EXTRA_DTYPES):
        tag = "flux_dev_1024_%s_b8_config_c5" % {"bfloat16": "bf16"}.get(dt_name, dt_name.replace("-", "_"))
        if time.time() - t_start > budget_s:
            out[tag] = "skipped: budget"
            continue
        try:
            from oracle.flux_ref import flops_per_image, latent_image_ids, ARCH_FLUX_DEV      # FLOP model + id helper only (not measured)
            cfg = dict(FLUX_CONFIGS["flux"])
            net = NativeFluxTransformer(cfg, device=dev, compute_dtype=dt_name)
            t0 = time.time(); net.init_synthetic(seed=0); torch.cuda.synchronize(); t_w = time.time() - t0
            B, grid, T = 8, 64, 512
            S = grid * grid
            g = torch.Generator(device=dev).manual_seed(1)
            x = torch.randn(B, S, cfg["in_channels"], generator=g, device=dev).half()
            enc = torch.randn(1, T, cfg["joint_attention_dim"], generator=g, device=dev).half().expand(B, -1, -1).contiguous()
            pooled = torch.randn(1, cfg["pooled_projection_dim"], generator=g, device=dev).half().expand(B, -1).contiguous()
            ts = torch.full((B,), 0.1, device=dev); gd = torch.full((B,), 1.0, device=dev)
            img_ids = latent_image_ids(grid, grid).to(dev); txt_ids = torch.zeros(T, 3, device=dev)
            nl = cfg["num_layers"] + cfg["num_single_layers"]
            picks = sorted({min(nl - 1, max(0, int(nl * f))) for f in (0.2, 0.4, 0.6, 0.8)})
            ids = [f"vit-block{i}-out" for i in picks] + [f"vit-block{i}-q" for i in picks]
            step = lambda: net.forward_raw(x, enc, pooled, ts, img_ids, txt_ids, guidance=gd, hook_ids=ids, grid=(grid, grid))
            steps = 2
            dt = _time_steps(step, steps, 2 if dt_name == "bfloat16" else 1)
            fl_img = flops_per_image(dict(ARCH_FLUX_DEV), S, T)
            out[tag] = {"images_per_s": round(B * steps / dt, 2), "ms_per_step": round(1e3 * dt / steps, 1), "hooks": len(ids),
                        "model_tflops_per_s": round(B * steps / dt * fl_img / 1e12, 1), "weights_init_s": round(t_w, 1),
                        "steps": steps, "dtype": dt_name,
                        "worst_hook_error_full_depth": {"bfloat16": "3.4e-3", "bfloat16x2": "1.8e-4", "fp8-mx": "7.5e-2 (opt-in, lower precision)",
                                                        "float16": "4.6e-4", "auto": "<= 8.5e-4 asserted (the FeatureExtractor default)"}.get(dt_name, "see tests/test_gpu_fullsize.py")}
            del net, step
            torch.cuda.empty_cache()
        except Exception as e:
            out[tag.replace("config_c5", "error")] = repr(e)[:300]
    # ---- PixArt-Sigma-XL-2 1024^2 (28 blocks, 4096 image + 300 caption tokens), batch 4: SURVEY §8(f) rank 4 ----
    if time.time() - t_start < budget_s:
        try:
            from components.native import NativePixArtTransformer, PIXART_CONFIGS
            from oracle.pixart_ref import ARCH_PIXART_SIGMA, flops_per_image as pix_flops       # FLOP model only
            net = NativePixArtTransformer(PIXART_CONFIGS["pixart-sigma"], device=dev).init_synthetic(0)
            B, T = 4, 300
            g = torch.Generator(device=dev).manual_seed(1)
            x = torch.randn(B, 4, 128, 128, device=dev, generator=g).half()
            enc = torch.randn(B, T, 4096, device=dev, generator=g).half()
            mask = (torch.arange(T, device=dev)[None] < 120).expand(B, T).to(torch.int64)
            tt = torch.full((B,), 100.0, device=dev)
            ids = [f"vit-block{i}-out" for i in (6, 13, 20, 27)]
            step = lambda: net.forward_raw(x, enc, tt, mask, hook_ids=ids)
            steps = 3
            dt = _time_steps(step, steps, 2)
            fl = pix_flops(ARCH_PIXART_SIGMA, 4096, T)
            out["pixart_sigma_1024_b4"] = {"images_per_s": round(B * steps / dt, 2), "ms_per_step": round(1e3 * dt / steps, 2), "hooks": len(ids),
                                           "model_tflops_per_s": round(B * steps / dt * fl / 1e12, 1), "steps": steps, "dtype": "f16"}
            del net, step
            torch.cuda.empty_cache()
        except Exception as e:
            out["pixart_error"] = repr(e)[:300]
    else:
        out["pixart_sigma_1024_b4"] = "skipped: budget"
    # ---- `vae-out`: scheduler step + AutoencoderKL decoder at 1024^2, batch 16 (reference diffusion_feature.py:477-485) ----
    if time.time() - t_start < budget_s:
        try:
            from components.native import NativeVAEDecoder, VAE_CONFIGS
            from oracle.vae_ref import ARCH_SD_VAE, dec_flops_per_image                        # FLOP model only
            dec = NativeVAEDecoder(VAE_CONFIGS["sd"], device=dev).init_synthetic(1)
            B = 16
            g = torch.Generator(device=dev).manual_seed(2)
            lat = torch.randn(B, 4, 128, 128, device=dev, generator=g).half(); npred = torch.randn(B, 4, 128, 128, device=dev, generator=g).half()
            step = lambda: dec.decode(lat, npred, c_sample=1.0, c_eps=-0.4, scaling_factor=0.13025)
            steps = 2
            dt = _time_steps(step, steps, 1)
            fl = dec_flops_per_image(ARCH_SD_VAE, 1024)
            out["vae_out_decode_1024_b16"] = {"images_per_s": round(B * steps / dt, 2), "ms_per_batch": round(1e3 * dt / steps, 1),
                                              "model_tflops_per_s": round(B * steps / dt * fl / 1e12, 1), "steps": steps, "dtype": "f16 operands, fp32 stream"}
            del dec, step
            torch.cuda.empty_cache()
        except Exception as e:
            out["vae_out_error"] = repr(e)[:300]
    else:
        out["vae_out_decode_1024_b16"] = "skipped: budget"
    out["seconds"] = round(time.time() - t_start, 1)
    return out


# ---- end to end: images -> hooks (-> .npy) ---------------------------------------------------------------------------------------------
def e2e_block(dev, version, B, img, practical_ids, steps=4):
    """FeatureExtractor.extract(image_type='tensors') on synthetic images resident in HBM: VAE encode + sample + noise-add, scheduler
    scaling, UNet forward, hooks — the whole product call; then the same loop with the CLI's D2H + np.save stage one batch behind."""
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if root not in sys.path:
        sys.path.insert(0, root)
    import diffusion_feature
    old = os.environ.get("GDF_SYNTHETIC_WEIGHTS")
    os.environ["GDF_SYNTHETIC_WEIGHTS"] = "1"
    try:
        df = diffusion_feature.FeatureExtractor({k: True for k in practical_ids}, version, device=str(dev), img_size=img)
    finally:
        if old is None:
            os.environ.pop("GDF_SYNTHETIC_WEIGHTS", None)
        else:
            os.environ["GDF_SYNTHETIC_WEIGHTS"] = old
    prompts = df.encode_prompt("a photo of a cat")
    g = torch.Generator(device=dev).manual_seed(3)
    imgs = (torch.rand(B, 3, img, img, generator=g, device=dev) * 2 - 1).half()
    step = lambda: df.extract(prompts, B, imgs, image_type="tensors", t=100)
    with torch.no_grad():
        dt = _time_steps(step, steps, 3)
        out = {"extract_images_per_s": round(B * steps / dt, 2), "extract_ms_per_batch": round(1e3 * dt / steps, 2),
               "stages": "VAE encode (AutoencoderKL, libgdf) + sample + noise-add -> scale_model_input -> UNet forward + hooks; image tensors resident in HBM"}
        # opt-in early exit (FeatureExtractor(early_exit=True)): the forward stops after the last requested layer — same features, the reference
        # itself always runs (and discards) the rest of the forward; NOT the headline and not the default
        df.pipe.unet.early_exit = True
        dte = _time_steps(step, steps, 3)
        df.pipe.unet.early_exit = False
        out["extract_early_exit_opt_in_images_per_s"] = round(B * steps / dte, 2)
        # the VAE stage alone (same call the extractor makes)
        lt = torch.full((B,), 100, device=dev)
        enc = lambda: df.pipe.prepare_latents(imgs, lt, 1, B, torch.float16, dev)
        dtv = _time_steps(enc, steps, 2)
        out["vae_encode_ms_per_batch"] = round(1e3 * dtv / steps, 2)
        out["vae_encode_images_per_s"] = round(B * steps / dtv, 2)
        # + the CLI's output stage: fp16 D2H into pinned memory on a side stream + np.save per (layer, image), one batch behind the GPU
        import argparse
        import extract_feature as cli
        tmp = tempfile.mkdtemp(prefix="gdf_e2e_", dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
        try:
            wargs = argparse.Namespace(output_dir=tmp, aggregate_output=False, sample_name_first=False)
            w = cli.HostWriter(wargs)
            def step_w(i=[0]):
                feats = step()
                w.submit(feats, [f"img{i[0]}_{j}" for j in range(B)])
                i[0] += 1
            for _ in range(2):
                step_w()
            w.flush(); torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(steps):
                step_w()
            w.flush(); torch.cuda.synchronize()
            dtw = time.perf_counter() - t0
            nbytes = sum(os.path.getsize(os.path.join(dp, f)) for dp, _, fs in os.walk(tmp) for f in fs)
            out["with_npy_output_images_per_s"] = round(B * steps / dtw, 2)
            out["npy_mb_per_image"] = round(nbytes / (B * (steps + 2)) / 1e6, 2)
            out["npy_note"] = "extract_feature.HostWriter (pinned D2H on a side stream, np.save per layer and image to a tmpfs directory on a writer thread, at most two batches behind)"
        finally:
            shutil.rmtree(tmp, ignore_errors=True)
        # the CLI's WHOLE loop from image files: JPEG decode + resize + normalise (host) -> VAE encode + UNet + hooks -> D2H -> .npy, with the
        # reference's serial input side (load the batch, then extract) and with the loader threads of round 5 (extract_feature.BatchLoader)
        try:
            out["cli_from_jpeg_files"] = _cli_from_files(df, prompts, B, img, cli, argparse)
        except Exception as e:
            out["cli_from_jpeg_files"] = {"error": repr(e)[:200]}
    del df
    torch.cuda.empty_cache()
    return out


def _cli_from_files(df, prompts, B, img, cli, argparse, n_images=128, n_serial=32):
    import glob
    import numpy as np
    from PIL import Image
    tmp = tempfile.mkdtemp(prefix="gdf_cli_", dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
    try:
        rs = np.random.RandomState(0)
        base = (rs.rand(160, 160, 3) * 255).astype(np.uint8)
        for i in range(n_images):
            Image.fromarray(np.roll(base, i, 0)).resize((1280, 960), Image.BICUBIC).save(os.path.join(tmp, f"img{i:04d}.jpg"), quality=92)
        paths = sorted(glob.glob(os.path.join(tmp, "*.jpg")))
        threads = min(32, max(2, (os.cpu_count() or 2) // 2))

        def run(thr, tag, paths=paths):
            w = cli.HostWriter(argparse.Namespace(output_dir=os.path.join(tmp, tag), aggregate_output=False, sample_name_first=False))
            starts = list(range(0, len(paths), B))
            loader = cli.BatchLoader(paths, starts, len(paths), B, df.preprocess_image, thr) if thr > 0 else None
            torch.cuda.synchronize(); t0 = time.perf_counter()
            with torch.no_grad():
                for i in starts:
                    chunk = paths[i:i + B]
                    if loader is not None:
                        feats = df.extract(prompts, len(chunk), loader.get(i), image_type="tensors", t=100)
                        loader.done(i)
                    else:
                        feats = df.extract(prompts, len(chunk), [Image.open(p) for p in chunk], t=100)
                    w.submit(feats, [f"train{i + j}" for j in range(len(chunk))])
                    del feats
            if loader is not None:
                loader.close()
            w.close(); torch.cuda.synchronize()
            return time.perf_counter() - t0
        run(threads, "warm", paths[:2 * B])
        dt_t = run(threads, "thr")
        old_env = os.environ.get("GDF_PREPROCESS_THREADS")
        os.environ["GDF_PREPROCESS_THREADS"] = "0"             # the reference's loop: one image after the other, then the GPU
        try:
            dt_s = run(0, "ser", paths[:n_serial])
        finally:
            if old_env is None:
                os.environ.pop("GDF_PREPROCESS_THREADS", None)
            else:
                os.environ["GDF_PREPROCESS_THREADS"] = old_env
        return {"images": n_images, "source": "1280x960 JPEG q92 files on tmpfs, resized to %dx%d" % (img, img), "host_cpus": os.cpu_count(),
                "loader_threads": threads, "images_per_s": round(n_images / dt_t, 2), "serial_input_loop_images_per_s": round(n_serial / dt_s, 2), "serial_input_loop_images": n_serial,
                "note": "extract_feature.py's loop: BatchLoader (decode / resize / normalise on threads, pinned fp16 batch buffers) -> FeatureExtractor.extract -> "
                        "HostWriter (pinned D2H, np.save on threads); `serial_input_loop` = the reference's order (load the batch, then extract), same files bit for bit"}
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
