"""Data-parallel plumbing for the hot path: one process per GPU, images sharded, weights broadcast once.

The reference has no distributed code; its only multi-GPU mode is one whole model per GPU in
Python threads (correspondence/correspondence/aggregation_network.py:67-95).  Here every rank holds identical
weights (rank 0's, broadcast at init over RCCL/xGMI — or gloo in the CPU tests) and processes its own slice of
the image batch; there is NO collective in the hot loop.
"""
import torch
import torch.distributed as dist

BUCKET_ELEMS = 1 << 28          # 512 MiB of fp16 per broadcast: few, large collectives


def shard_range(n_items, rank, world):
    """Contiguous slice [lo, hi) of `n_items` images owned by `rank` (sizes differ by at most one)."""
    base, rem = divmod(n_items, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def broadcast_state_dict(shapes, make_tensor, consume, device, src=0, bucket_elems=BUCKET_ELEMS, dtype=torch.float16):
    """Stream a state dict from rank `src` to all ranks in flat buckets.

    shapes: ordered {name: shape};  make_tensor(name, shape) -> tensor (called on `src` only);
    consume(dict name -> tensor view) is called on EVERY rank once per bucket (views die with the bucket)."""
    rank = dist.get_rank() if dist.is_initialized() else 0
    names = list(shapes)
    i = 0
    while i < len(names):
        j, tot = i, 0
        while j < len(names) and (tot == 0 or tot + _numel(shapes[names[j]]) <= bucket_elems):
            tot += _numel(shapes[names[j]])
            j += 1
        flat = torch.empty(tot, dtype=dtype, device=device)
        if rank == src:
            off = 0
            for n in names[i:j]:
                k = _numel(shapes[n])
                flat[off:off + k] = make_tensor(n, shapes[n]).reshape(-1).to(dtype)
                off += k
        if dist.is_initialized() and dist.get_world_size() > 1:
            dist.broadcast(flat, src=src)
        off, sd = 0, {}
        for n in names[i:j]:
            k = _numel(shapes[n])
            sd[n] = flat[off:off + k].view(shapes[n])
            off += k
        consume(sd)
        i = j


def rank_world():
    """(rank, world) of the initialised process group, (0, 1) without one."""
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


def broadcast_model_weights(model, src=0, chunk_bytes=1 << 29):
    """Broadcast the flat device weight arena of a native model (components.native `weight_blob`) from rank `src` in
    512 MiB pieces (few, large collectives: ring broadcast over xGMI is per-link bound); receivers mark the model ready.
    The arena is already in the kernels' layout, so no rank but `src` reads or re-lays-out a checkpoint."""
    rank, world = rank_world()
    if world == 1:
        return model
    blob = model.weight_blob()
    via_host = dist.get_backend() == "gloo"            # CPU-side test backend (two ranks on one GPU): stage through host memory
    for off in range(0, blob.numel(), chunk_bytes):
        piece = blob[off:off + chunk_bytes]
        if via_host:
            h = piece.cpu() if rank == src else torch.empty(piece.shape, dtype=piece.dtype)
            dist.broadcast(h, src=src)
            if rank != src:
                piece.copy_(h)
        else:
            dist.broadcast(piece, src=src)
    if rank != src:
        model.set_ready()
    return model


def synthetic_param(name, shape, gen, device):
    """Seeded synthetic weight (no checkpoints offline): W ~ N(0,1/fan_in), bias 0.05 N, gamma 1+0.1 N, beta 0.1 N."""
    is_norm = ".norm" in name or name.startswith("conv_norm_out")
    t = torch.randn(shape, generator=gen, device=device, dtype=torch.float32)
    if name.endswith(".weight") and not is_norm:
        fan = 1
        for s in shape[1:]:
            fan *= s
        t.mul_(fan ** -0.5)
    elif name.endswith(".weight"):
        t.mul_(0.1).add_(1.0)
    elif is_norm:
        t.mul_(0.1)
    else:
        t.mul_(0.05)
    return t.half()


def _numel(shape):
    n = 1
    for s in shape:
        n *= s
    return n
