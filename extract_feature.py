#!/usr/bin/env python3
"""Feature-extraction CLI for the MI355X-native extractor — same flags and on-disk layout as the reference's
extract_feature.py (/root/reference/extract_feature.py:18-43 flags, :113-148 output stage, figures/output_format.jpg):

  per-layer mode      <output_dir>/<layer_id>/<name>.npy            (C,H,W) fp16        [default]
  --sample_name_first <output_dir>/<name>/<layer_id>.npy
  --aggregate_output  <output_dir>/<name>.npy                       all layers nearest-resized to the largest H,W and
                                                                    concatenated over channels
  <name> = original file stem (--use_original_filename; parent-dir/stem with --nested_input_dir) or <split><index>.

The reference's own script also runs unchanged against this package (it only needs `import diffusion_feature`);
this version overlaps the device->host copies of batch i with the extraction of batch i+1 through pinned buffers.

Data-parallel (one process per GPU, no collective in the loop):
    python3 extract_feature.py --gpus 8 ...          (starts its 8 ranks itself: components/dist.py self_launch)
    python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 extract_feature.py ...   (equivalent)
rank r extracts the contiguous slice shard_range(len(images), r, W) of the sorted image list and writes its own files
(<split><GLOBAL index> names, so the output directory is identical to a single-process run); rank 0 alone reads / generates
the denoiser weights and broadcasts the flat device arena once over RCCL (components/dist.py).
"""
import argparse
import glob
import json
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "generic-diffusion-feature_amd"))
import diffusion_feature  # noqa: E402


def parse_args(argv=None):
    p = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    # model / extractor settings (constructor keywords of FeatureExtractor)
    p.add_argument('--layer', type=str, help="layer-selection json ({layer_id: bool})")
    p.add_argument('--version', type=str, default='xl')
    p.add_argument('--dtype', type=str, default='float16', choices=('float16', 'float32'))
    p.add_argument('--offline_lora', type=str, default=None)
    p.add_argument('--offline_lora_filename', type=str, default=None)
    p.add_argument('--feature_resize', type=int, default=1)
    p.add_argument('--control', type=str, nargs='+', default=None)
    p.add_argument('--attention', type=str, nargs='+', default=None,
                   choices=('down_cross', 'mid_cross', 'up_cross', 'down_self', 'mid_self', 'up_self'))
    p.add_argument('--img_size', type=int, default=1024)
    # extraction settings
    p.add_argument('--batch_size', '-b', type=int, default=2)
    p.add_argument('--t', type=int, help='timestep at which features are extracted')
    p.add_argument('--denoising_from', type=int, default=None)
    p.add_argument('--use_ddim_inversion', action='store_true')
    # io settings
    p.add_argument('--input_dir', type=str, default=None, help='glob pattern of the input images')
    p.add_argument('--nested_input_dir', action='store_true')
    p.add_argument('--prompt_file', type=str, default='prompt.txt')
    p.add_argument('--output_dir', type=str, default='./output/')
    p.add_argument('--aggregate_output', action='store_true')
    p.add_argument('--use_original_filename', action='store_true')
    p.add_argument('--split', type=str, default='train')
    p.add_argument('--sample_name_first', action='store_true')
    p.add_argument('--show_all_layers', action='store_true')
    p.add_argument('--precise', nargs='?', const='precise', default=None,
                   help="native extension (not in the reference CLI), UNet versions only: operand plan.  Default (flag absent) = 'auto': the cheapest "
                        "plan that keeps every requested layer within 1e-3 of the fp32 reference (plain fp16 operands, the selective split or the "
                        "full split).  --precise = the full split (every feature <= 5e-4, ~1.6x the time); --precise selective | plain | "
                        "a class list such as stream,attn_out")
    p.add_argument('--early_exit', action='store_true',
                   help='native extension (not in the reference CLI): stop the denoiser forward after the last requested layer (same files, less work)')
    p.add_argument('--loader_threads', type=int, default=-1,
                   help='native extension (not in the reference CLI): threads that decode + resize + normalise the NEXT batches while the GPU works on the current '
                        'one (PIL and numpy release the GIL).  -1 = min(32, half of this rank\'s share of the host CPUs); 0 = the reference\'s serial loop (load, then extract)')
    p.add_argument('--seed', type=int, default=None,
                   help="native extension: seed of the VAE-sampling / add-noise draws.  Absent = the reference's behaviour (the global torch RNG: "
                        "run- and partition-dependent with a real diffusers pipeline).  Given: the generator is re-seeded with seed + <index of the "
                        "batch's first image> before every batch, so the features of an image depend only on (seed, batch start) — identical for "
                        "any --gpus N whose shards are whole batches")
    p.add_argument('--gpus', type=int, default=1,
                   help='native extension (not in the reference CLI): data-parallel over N GPUs of this node.  Started as a plain process '
                        '(`python3 extract_feature.py --gpus 8 ...`) the script starts its N ranks itself; under torchrun it must equal WORLD_SIZE')
    return p.parse_args(argv)


def sample_name(path, nested):
    stem = os.path.splitext(os.path.basename(path))[0]
    return os.path.join(os.path.basename(os.path.dirname(path)), stem) if nested else stem


class HostWriter:
    """Device -> pinned host -> .npy, behind the GPU: the D2H copies run on a side stream, the np.save calls on a writer thread (round 5: the
    main thread used to write the previous batch's files itself — 15.7 MB per SDXL image through single-threaded np.save — before it could
    launch the next extract: 68 vs 74 img/s end to end).  At most `depth` batches are pending; flush() waits for all of them and re-raises
    a writer error."""

    def __init__(self, args, depth=2):
        import queue
        import threading
        self.args = args
        self.stream = torch.cuda.Stream() if torch.cuda.is_available() else None
        self.q = queue.Queue(maxsize=depth)
        self.err = None
        from concurrent.futures import ThreadPoolExecutor
        self.savers = ThreadPoolExecutor(max_workers=4, thread_name_prefix="gdf-save") if (os.cpu_count() or 1) >= 8 else None
        self.thread = threading.Thread(target=self._work, daemon=True)
        self.thread.start()

    def _work(self):
        while True:
            item = self.q.get()
            try:
                if item is None:
                    return
                if self.err is None:
                    self._write(*item)
            except BaseException as e:               # surfaced in the main thread by submit() / flush()
                self.err = e
            finally:
                self.q.task_done()

    def _check(self):
        if self.err is not None:
            e, self.err = self.err, None
            raise e

    def submit(self, feats, names):
        self._check()
        host = {}
        if self.stream is not None:
            self.stream.wait_stream(torch.cuda.current_stream())
        ctx = torch.cuda.stream(self.stream) if self.stream is not None else _null()
        originals = [v for v in feats.values() if torch.is_tensor(v)]      # what the extractor handed out (aggregation rebinds `feats`)
        with ctx:
            if self.args.aggregate_output:                      # reference :113-125; device tensors: resize_concat_kernel
                from components.postproc import resize_concat
                feats = {None: resize_concat(list(feats.values()))}
            for k, v in feats.items():
                v = v.detach()
                if v.is_cuda:
                    buf = torch.empty(v.shape, dtype=v.dtype, pin_memory=True)
                    buf.copy_(v, non_blocking=True)
                    host[k] = buf
                else:
                    host[k] = v
        ev = torch.cuda.Event() if self.stream is not None else None
        if ev is not None:
            ev.record(self.stream)
            # the D2H copies (and resize_concat) read the hook buffers on THIS stream: announce it, so that the extractor may recycle the
            # buffers as soon as `feats` is dropped (components/native.py release_after — the record_stream() of the native hook buffers);
            # tensors that come from the caching allocator instead (pooled / aggregated features, the concat result) get record_stream()
            seen = originals + [v for v in feats.values() if torch.is_tensor(v)]
            try:
                from components.native import release_after
                rest = release_after(seen, self.stream)
            except ImportError:
                rest = [v for v in seen if v.is_cuda]
            for v in rest:
                v.record_stream(self.stream)
        self.q.put((host, names, ev))                            # blocks while `depth` batches are still being written

    def flush(self):
        self.q.join()
        self._check()

    def close(self):
        self.q.join()
        self.q.put(None)
        self.thread.join()
        self._check()

    def _write(self, host, names, ev):
        if ev is not None:
            ev.synchronize()
        a = self.args
        jobs = []
        for j, name in enumerate(names):
            for k, v in host.items():
                if k is None:                                       # aggregated: <output_dir>/<name>.npy
                    path = os.path.join(a.output_dir, name)
                elif a.sample_name_first:
                    path = os.path.join(a.output_dir, name, k)
                else:
                    path = os.path.join(a.output_dir, k, name)
                os.makedirs(os.path.dirname(path) or '.', exist_ok=True)
                jobs.append((path, v[j].numpy()))
        # np.save of a C-contiguous array is one header + one write(): it releases the GIL, a few writers in parallel keep up with
        # 1 GB/s of features (SDXL practical set: 15.7 MB per image at ~70 images/s)
        if self.savers is None or len(jobs) < 4:
            for path, arr in jobs:
                np.save(path, arr)
        else:
            list(self.savers.map(lambda pa: np.save(pa[0], pa[1]), jobs))


class BatchLoader:
    """Input side of the CLI, ahead of the GPU: a thread pool opens, resizes and normalises the images of the next `depth` batches (Image.open +
    FeatureExtractor.preprocess_image per image: 20-45 ms of host time per 1024^2 JPEG, 0.3-0.7 s per batch of 16 against 0.22 s of GPU work — the
    reference's serial loop, and this CLI until round 5, leave the GPU idle for it).  get(i) returns the (B, 3, S, S) tensor of the batch that
    starts at image i, in order; exceptions of a worker surface there.
    With a GPU the workers write fp16 straight into one of `depth + 2` rotating PINNED batch buffers (the VAE stage consumes fp16; the serial path's
    `copy_` performs the same round-to-nearest conversion), so the upload is one asynchronous DMA instead of a pageable fp32 copy; done(i) records the
    event after which batch i's buffer may be overwritten."""

    def __init__(self, paths, starts, hi, batch_size, preprocess, threads, depth=2):
        from concurrent.futures import ThreadPoolExecutor
        self.paths, self.starts, self.hi, self.bs, self.pre, self.depth = paths, list(starts), hi, batch_size, preprocess, depth
        self.pool = ThreadPoolExecutor(max_workers=max(1, threads), thread_name_prefix="gdf-loader")
        import threading
        self.lock = threading.Lock()
        self.pinned = torch.cuda.is_available()
        # the CUDA device is per THREAD and a new thread starts on device 0: the workers allocate their pinned buffers under THIS rank's device,
        # or every rank of an N-GPU job would open a context on GPU 0 for its host allocator
        self.dev = torch.cuda.current_device() if self.pinned else None
        # depth + 2 buffers: the batch submitted during get(i) reuses the buffer of batch i - 2, whose forward has long finished (with depth + 1
        # it would be batch i - 1's, i.e. the forward that has just been queued: the main thread would wait for it)
        self.nbuf = depth + 2
        self.bufs, self.events, self.slot_of = [None] * self.nbuf, [None] * self.nbuf, {}
        self.futs = {}
        self.next = 0
        for _ in range(depth):
            self._submit()

    def _one(self, path, slot, j):
        from PIL import Image
        with Image.open(path) as im:
            x = self.pre(im)                                        # (1, 3, S, S) float, the serial path's own function
        if slot is None:
            return x
        buf = self.bufs[slot]
        if buf is None or tuple(buf.shape[1:]) != tuple(x.shape[1:]):
            with self.lock:
                buf = self.bufs[slot]
                if buf is None or tuple(buf.shape[1:]) != tuple(x.shape[1:]):
                    with torch.cuda.device(self.dev):
                        buf = self.bufs[slot] = torch.empty((self.bs,) + tuple(x.shape[1:]), dtype=torch.float16, pin_memory=True)
        buf[j].copy_(x[0])
        return None

    def _submit(self):
        if self.next < len(self.starts):
            k = self.next
            i = self.starts[k]
            self.next += 1
            slot = k % self.nbuf if self.pinned else None
            if slot is not None and self.events[slot] is not None:
                self.events[slot].synchronize()                     # the batch that used this buffer (depth + 2 batches ago) has been consumed
            self.slot_of[i] = slot
            self.futs[i] = [self.pool.submit(self._one, p, slot, j) for j, p in enumerate(self.paths[i:min(i + self.bs, self.hi)])]

    def get(self, i):
        fs = self.futs.pop(i)
        res = [f.result() for f in fs]
        slot = self.slot_of[i]
        out = torch.concat(res, dim=0) if slot is None else self.bufs[slot][:len(res)]
        self._submit()
        return out

    def done(self, i):
        """call after the work that reads batch i has been queued on the current stream"""
        slot = self.slot_of.pop(i, None)
        if slot is not None:
            ev = torch.cuda.Event()
            ev.record()
            self.events[slot] = ev

    def close(self):
        self.pool.shutdown(wait=False, cancel_futures=True)


class _null:
    def __enter__(self):
        return self

    def __exit__(self, *a):
        return False


def init_data_parallel():
    """torchrun environment -> (rank, world, device string).  Must run before anything touches the GPU."""
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    from components import dist as D
    if world == 1 and not D.one_rank_group():     # (GDF_RCCL_ONE_RANK=1: test hook, the N-rank path in a one-rank RCCL group)
        return 0, 1, 'cuda'
    import torch.distributed as dist
    D.pin_rank_cores()                # this rank's share of the host cores (launch thread, loader threads, OpenMP pool), before any GPU call
    share = os.environ.get("GDF_SHARE_GPU", "0") == "1"              # test hook: every rank on cuda:0, gloo instead of RCCL
    local = 0 if share else int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local)
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29534")
    try:
        if share:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device(f"cuda:{local}"))
    except RuntimeError as e:
        if rank == 0 and ("EADDRINUSE" in str(e) or "address already in use" in str(e).lower()):
            raise SystemExit(D.RENDEZVOUS_BIND_FAILED)               # self_launch retries once on another port
        raise
    D.enable_weight_broadcast()       # every rank builds the same extractor below: rank 0 loads, the others receive the arena
    return rank, world, f"cuda:{local}"


def main(argv=None):
    from PIL import Image
    args = parse_args(argv)
    if args.precise is not None and (args.version == 'flux' or args.version.startswith('pixart')):
        raise SystemExit("--precise: split-operand plans exist for the UNet versions ('1-5', '2-1', 'xl', 'pgv2') only")
    # the front door for N ranks: a plain `python3 extract_feature.py --gpus N ...` starts one child process per GPU (this parent never
    # touches the GPU) and exits with their code; under torchrun (WORLD_SIZE set) the ranks arrive here directly
    from components import dist as D
    if D.needs_self_launch(args.gpus):
        if os.environ.get("GDF_SHARE_GPU", "0") != "1" and torch.cuda.device_count() < args.gpus:
            raise SystemExit(f"--gpus {args.gpus} needs {args.gpus} visible GPUs, torch.cuda.device_count() = {torch.cuda.device_count()}")
        raise SystemExit(D.self_launch(os.path.abspath(__file__), sys.argv[1:] if argv is None else list(argv), args.gpus))
    if args.gpus > 1 and int(os.environ.get("WORLD_SIZE", "1")) != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} does not match WORLD_SIZE={os.environ.get('WORLD_SIZE')}: one rank per GPU")
    rank, world, device = init_data_parallel()
    os.makedirs(args.output_dir, exist_ok=True)
    if rank == 0:
        print(f'Run folder: {args.output_dir}')
    if args.show_all_layers:
        args.layer = None
    df = diffusion_feature.FeatureExtractor(
        args.layer, args.version, device=device, dtype=args.dtype, offline_lora=args.offline_lora,
        offline_lora_filename=args.offline_lora_filename, feature_resize=args.feature_resize, control=args.control,
        attention=args.attention, img_size=args.img_size,
        precise=None if args.precise is None else (False if args.precise == 'plain' else args.precise), early_exit=args.early_exit)

    paths = sorted(glob.glob(args.input_dir, recursive=True))
    lo, hi = 0, len(paths)
    if world > 1:
        from components.dist import shard_range
        lo, hi = shard_range(len(paths), rank, world)
    with open(args.prompt_file, 'r') as f:
        prompt_text = f.read()
    if rank == 0:
        print('prompt:', prompt_text)
    # reference extract_feature.py:81-82: flux / hunyuan pipelines take the raw prompt text
    prompts = prompt_text if args.version in ('flux', 'hunyuan') else df.encode_prompt(prompt_text)

    writer = HostWriter(args)
    starts = list(range(lo, hi, args.batch_size))
    # default: up to 32 loader threads, but never more than half of this rank's share of the host CPUs (8 ranks on a 256-CPU node: 16 each)
    ranks_here = max(1, int(os.environ.get("LOCAL_WORLD_SIZE", "1") or 1))
    n_thr = min(32, max(2, (os.cpu_count() or 2) // (2 * ranks_here))) if args.loader_threads < 0 else args.loader_threads
    # flux / hunyuan pipelines take PIL images (reference :246-254); the UNet / PixArt versions go through df.preprocess_image, which is what the
    # loader threads run — the SAME function the serial path calls, so the latents are bit-identical either way
    prefetch = n_thr > 0 and args.version not in ('flux', 'hunyuan') and not args.show_all_layers
    loader = BatchLoader(paths, starts, hi, args.batch_size, df.preprocess_image, n_thr) if prefetch else None
    ok = False
    try:
        with torch.no_grad():
            for i in starts:
                chunk = paths[i:min(i + args.batch_size, hi)]
                if args.seed is not None:
                    torch.manual_seed(args.seed + i)
                if loader is not None:
                    feats = df.extract(prompts, len(chunk), loader.get(i), image_type='tensors', t=args.t, denoising_from=args.denoising_from,
                                       use_control=args.control is not None, use_ddim_inversion=args.use_ddim_inversion)
                    loader.done(i)
                else:
                    images = [Image.open(p) for p in chunk]
                    feats = df.extract(prompts, len(images), images, t=args.t, denoising_from=args.denoising_from,
                                       use_control=args.control is not None, use_ddim_inversion=args.use_ddim_inversion)
                if args.show_all_layers:                               # dump the id list and stop (reference :103-110)
                    for k, v in feats.items():
                        print(k, tuple(v[0].shape))
                    with open('layer_record.json', 'w') as f:
                        json.dump({k: True for k in feats}, f)
                    return
                names = [sample_name(p, args.nested_input_dir) if args.use_original_filename else f'{args.split}{i + j}'
                         for j, p in enumerate(chunk)]
                writer.submit(feats, names)
                if rank == 0:
                    print(f'{min(i + len(chunk), hi) - lo}/{hi - lo}', end='\r')
        ok = True
    finally:
        # success, an exception in extract / the loader (a broken image file), or the --show_all_layers early return: the loader's thread pool and
        # pinned batch buffers and the writer thread are always released.  On the error path the writer's own (later) exception must not replace
        # the one that is already propagating.
        if loader is not None:
            loader.close()
        try:
            writer.close()
        except Exception:
            if ok:
                raise
    if args.show_all_layers:
        return
    if world > 1:
        import torch.distributed as dist
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print()


if __name__ == '__main__':
    main()
