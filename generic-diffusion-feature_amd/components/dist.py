"""Data-parallel plumbing for the hot path: one process per GPU, images sharded, weights broadcast once.

The reference has no distributed code; its only multi-GPU mode is one whole model per GPU in
Python threads (correspondence/correspondence/aggregation_network.py:67-95).  Here every rank holds identical
weights (rank 0's, broadcast at init over RCCL/xGMI — or gloo in the CPU tests) and processes its own slice of
the image batch; there is NO collective in the hot loop.
"""
import torch
import torch.distributed as dist

import os

# Data-parallel weight sharing is OPT-IN: a host program that merely has a torch.distributed group initialised (a DDP trainer using
# the extractor as a frozen backbone on some ranks, one extractor per thread, ...) must not be dragged into collectives by the
# model constructors.  extract_feature.py / bench.py (the launches that build the same model on EVERY rank, in the same order)
# call enable_weight_broadcast(); GDF_DP_BROADCAST=1 does the same from the environment.
_broadcast_enabled = os.environ.get("GDF_DP_BROADCAST", "0") not in ("", "0")


def enable_weight_broadcast(on=True):
    global _broadcast_enabled
    _broadcast_enabled = bool(on)


def weight_broadcast_enabled():
    return _broadcast_enabled and rank_world()[1] > 1


def broadcast_object(obj, src=0):
    """Small picklable object (a config dict) from rank `src` to every rank; identity without a process group."""
    rank, world = rank_world()
    if world == 1:
        return obj
    box = [obj if rank == src else None]
    dist.broadcast_object_list(box, src=src)
    return box[0]


def shard_range(n_items, rank, world):
    """Contiguous slice [lo, hi) of `n_items` images owned by `rank` (sizes differ by at most one)."""
    base, rem = divmod(n_items, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


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
    # every rank must hold an arena of the same size (same architecture descriptor) before any piece moves
    n = torch.tensor([blob.numel(), -blob.numel()], dtype=torch.int64, device="cpu" if via_host else blob.device)
    dist.all_reduce(n, op=dist.ReduceOp.MAX)
    if int(n[0]) != blob.numel() or int(-n[1]) != blob.numel():
        raise RuntimeError(f"weight arenas differ across ranks ({blob.numel()} bytes here, {int(-n[1])}..{int(n[0])} in the group): "
                           "the ranks did not build the same model")
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


