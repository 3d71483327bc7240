"""N>1 path on CPU: world_size-2 gloo — bucketed weight broadcast reproduces rank 0's state dict on every rank,
and the image sharding covers the batch exactly once with no data-path collective."""
import os
import sys

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "generic-diffusion-feature_amd"))


def _worker(rank, world, port, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from components import dist as D
    shapes = {"a.weight": (7, 5, 3, 3), "a.bias": (7,), "b.norm.weight": (33,), "c.weight": (129, 65), "d.bias": (3,)}
    gen = torch.Generator().manual_seed(123 + rank)          # different seeds: only rank 0's values may survive
    got = {}
    D.broadcast_state_dict(shapes, lambda n, s: D.synthetic_param(n, s, gen, "cpu"),
                           lambda sd: got.update({k: v.clone() for k, v in sd.items()}), "cpu", bucket_elems=400)
    lo, hi = D.shard_range(11, rank, world)

    class Arena:                                             # stand-in for a native model's flat device weight arena
        def __init__(self):
            self.blob = torch.randint(0, 255, (1000,), dtype=torch.uint8, generator=torch.Generator().manual_seed(7 + rank))
            self.ready = rank == 0

        def weight_blob(self):
            return self.blob

        def set_ready(self):
            self.ready = True
    ar = Arena()
    D.broadcast_model_weights(ar, chunk_bytes=300)           # 4 pieces
    torch.save({"sd": got, "range": (lo, hi), "blob": ar.blob, "ready": ar.ready, "rw": D.rank_world()}, os.path.join(out, f"r{rank}.pt"))
    dist.destroy_process_group()


def test_broadcast_and_shard_gloo(tmp_path):
    world, port = 2, 29500 + os.getpid() % 2000
    mp.spawn(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    r0, r1 = torch.load(tmp_path / "r0.pt"), torch.load(tmp_path / "r1.pt")
    from components import dist as D
    gen = torch.Generator().manual_seed(123)
    for n, s in {"a.weight": (7, 5, 3, 3), "a.bias": (7,), "b.norm.weight": (33,), "c.weight": (129, 65), "d.bias": (3,)}.items():
        ref = D.synthetic_param(n, s, gen, "cpu")
        assert torch.equal(r0["sd"][n], ref) and torch.equal(r1["sd"][n], ref), n     # bit-exact on both ranks
    assert r0["range"] == (0, 6) and r1["range"] == (6, 11)
    # flat-arena broadcast (what extract_feature.py / bench.py use): rank 1 ends up with rank 0's bytes and is marked ready
    want = torch.randint(0, 255, (1000,), dtype=torch.uint8, generator=torch.Generator().manual_seed(7))
    assert torch.equal(r0["blob"], want) and torch.equal(r1["blob"], want) and r0["ready"] and r1["ready"]
    assert r0["rw"] == (0, 2) and r1["rw"] == (1, 2)


def test_shard_range_partitions():
    from components.dist import shard_range
    for n in (0, 1, 7, 16, 128, 129):
        for w in (1, 2, 3, 8):
            spans = [shard_range(n, r, w) for r in range(w)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(spans[i][1] == spans[i + 1][0] for i in range(w - 1))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1
