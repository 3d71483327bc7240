"""N>1 path on CPU: world_size-2 gloo — the flat weight-arena broadcast reproduces rank 0's bytes on every rank (opt-in, size
checked), the architecture descriptor travels from rank 0, and the image sharding covers the batch exactly once with no
data-path collective."""
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
    from components import models as M
    lo, hi = D.shard_range(11, rank, world)

    class Arena:                                             # stand-in for a native model's flat device weight arena
        def __init__(self, n=1000):
            self.blob = torch.randint(0, 255, (n,), dtype=torch.uint8, generator=torch.Generator().manual_seed(7 + rank))
            self.ready = rank == 0
            self.loaded = False

        def weight_blob(self):
            return self.blob

        def set_ready(self):
            self.ready = True
    ar = Arena()
    D.broadcast_model_weights(ar, chunk_bytes=300)           # 4 pieces
    # the architecture descriptor travels from rank 0 (only rank 0 has the loaded UNet module: components/models.py)
    cfg = D.broadcast_object(dict(block_out_channels=(320, 640), heads=(5, 10)) if rank == 0 else None)
    # models._fill: a process group alone does NOT drag a constructor into collectives (opt-in) ...
    solo = Arena()
    assert not D.weight_broadcast_enabled()
    M._fill(solo, lambda m: setattr(m, "loaded", True))
    solo_ok = solo.loaded and torch.equal(solo.blob, Arena().blob)
    # ... after enable_weight_broadcast() rank 0 loads and the others receive
    D.enable_weight_broadcast()
    dp = Arena()
    M._fill(dp, lambda m: setattr(m, "loaded", True))
    # arenas of different sizes (ranks built different models) are refused before any byte moves
    mismatch = None
    try:
        D.broadcast_model_weights(Arena(1000 + 8 * rank))
    except RuntimeError as e:
        mismatch = str(e)
    torch.save({"range": (lo, hi), "blob": ar.blob, "ready": ar.ready, "rw": D.rank_world(), "cfg": cfg, "solo_ok": solo_ok,
                "dp_loaded": dp.loaded, "dp_blob": dp.blob, "dp_ready": dp.ready, "mismatch": mismatch}, os.path.join(out, f"r{rank}.pt"))
    dist.destroy_process_group()


def test_broadcast_and_shard_gloo(tmp_path):
    world, port = 2, 29500 + os.getpid() % 2000
    mp.spawn(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    r0, r1 = torch.load(tmp_path / "r0.pt"), torch.load(tmp_path / "r1.pt")
    assert r0["range"] == (0, 6) and r1["range"] == (6, 11)
    # flat-arena broadcast (what extract_feature.py / bench.py use): rank 1 ends up with rank 0's bytes and is marked ready
    want = torch.randint(0, 255, (1000,), dtype=torch.uint8, generator=torch.Generator().manual_seed(7))
    assert torch.equal(r0["blob"], want) and torch.equal(r1["blob"], want) and r0["ready"] and r1["ready"]
    assert r0["rw"] == (0, 2) and r1["rw"] == (1, 2)
    assert r0["cfg"] == r1["cfg"] == dict(block_out_channels=(320, 640), heads=(5, 10))
    assert r0["solo_ok"] and r1["solo_ok"]                                   # no collective without the opt-in: both ranks loaded
    assert r0["dp_loaded"] and not r1["dp_loaded"]                           # opt-in: only rank 0 runs the loader
    assert torch.equal(r0["dp_blob"], want) and torch.equal(r1["dp_blob"], want) and r1["dp_ready"]
    assert r0["mismatch"] and r1["mismatch"] and "differ across ranks" in r1["mismatch"]


def test_config_from_diffusers_fills_constructor_defaults():
    """The raw unet/config.json of SD1.5 has neither `transformer_layers_per_block` nor `use_linear_projection`."""
    import types
    from components.native import config_from_diffusers, ARCH_CONFIGS
    raw15 = types.SimpleNamespace(in_channels=4, out_channels=4, block_out_channels=[320, 640, 1280, 1280], attention_head_dim=8,
                                  down_block_types=["CrossAttnDownBlock2D"] * 3 + ["DownBlock2D"], layers_per_block=2,
                                  cross_attention_dim=768)
    assert config_from_diffusers(raw15) == ARCH_CONFIGS["1-5"]
    xl = types.SimpleNamespace(in_channels=4, out_channels=4, block_out_channels=[320, 640, 1280], attention_head_dim=[5, 10, 20],
                               down_block_types=["DownBlock2D", "CrossAttnDownBlock2D", "CrossAttnDownBlock2D"], layers_per_block=2,
                               cross_attention_dim=2048, transformer_layers_per_block=[1, 2, 10], use_linear_projection=True,
                               addition_embed_type="text_time", addition_time_embed_dim=256, projection_class_embeddings_input_dim=2816)
    assert config_from_diffusers(xl) == ARCH_CONFIGS["xl"]


def test_shard_range_partitions():
    from components.dist import shard_range
    for n in (0, 1, 7, 16, 128, 129):
        for w in (1, 2, 3, 8):
            spans = [shard_range(n, r, w) for r in range(w)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(spans[i][1] == spans[i + 1][0] for i in range(w - 1))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1


_RANK_SCRIPT = r'''
import json, os, sys
sys.path.insert(0, sys.argv[1])
import torch.distributed as dist
from components import dist as D
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
assert os.environ["GDF_SELF_LAUNCHED"] == "1" and os.environ["LOCAL_RANK"] == str(rank) and os.environ["MASTER_ADDR"] == "127.0.0.1"
if len(sys.argv) > 2 and sys.argv[2] == "fail" and rank == 1:
    sys.exit(7)                                   # a rank that dies before the rendezvous: the job must come down, not hang
dist.init_process_group("gloo", rank=rank, world_size=world)
ev = D.group_evidence(None)
print("noise from rank", rank)                     # only rank 0's stdout is the command's stdout
if rank == 0:
    print(json.dumps(ev))
dist.barrier()
dist.destroy_process_group()
'''


def test_self_launch_front_door(tmp_path):
    """`python3 bench.py --gpus N` / `extract_feature.py --gpus N` start their N ranks themselves (components/dist.py self_launch):
    torchrun-compatible environment, rank 0 owns stdout, every rank answers the evidence all_reduce, a failing rank fails the job."""
    import json
    import subprocess
    pkg = os.path.join(ROOT, "generic-diffusion-feature_amd")
    script = tmp_path / "rank.py"
    script.write_text(_RANK_SCRIPT)
    drv = ("import sys; sys.path.insert(0, %r); from components import dist as D; "
           "assert D.needs_self_launch(3) and not D.needs_self_launch(1); "
           "sys.exit(D.self_launch(%r, sys.argv[1:], 3, timeout_s=120))" % (pkg, str(script)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, "-c", drv, pkg], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    out = [l for l in r.stdout.splitlines() if l.strip()]
    ev = json.loads([l for l in out if l.startswith("{")][0])
    assert ev["world_size"] == 3 and ev["ranks_seen"] == [0, 1, 2] and ev["backend"] == "gloo"
    assert "noise from rank 0" in r.stdout and "noise from rank 1" not in r.stdout and "noise from rank 1" in r.stderr
    r = subprocess.run([sys.executable, "-c", drv, pkg, "fail"], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 7 and "rank 1 exited with code 7" in r.stderr
    # inside a rank (WORLD_SIZE set) the front door stays shut
    r = subprocess.run([sys.executable, "-c", "import sys; sys.path.insert(0, %r); from components import dist as D; "
                        "sys.exit(int(D.needs_self_launch(4)))" % pkg], env=dict(env, WORLD_SIZE="4", RANK="0"), timeout=120)
    assert r.returncode == 0


def test_bench_front_door_refuses_without_gpus():
    """No GPU here: `python3 bench.py --gpus 2` must fail fast in the PARENT (device_count() < N) without starting ranks or touching a GPU."""
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "GDF_BENCH_SHARE_GPU")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1"], env=env, capture_output=True, text=True, timeout=300)
    if torch.cuda.device_count() < 2:
        assert r.returncode != 0 and "needs 2 visible GPUs" in r.stderr and "self_launch" not in r.stderr


def _one_rank_worker(rank, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), GDF_RCCL_ONE_RANK="1")
    from components import dist as D
    before = (D._grouped(), D.group_evidence()["backend"])             # hook set but no group yet: the collectives stay skipped
    dist.init_process_group("gloo", rank=0, world_size=1)

    class Arena:
        def __init__(self):
            self.blob = torch.arange(700, dtype=torch.uint8)
            self.ready = True

        def weight_blob(self):
            return self.blob

        def set_ready(self):
            self.ready = True
    ar = Arena()
    D.broadcast_model_weights(ar, chunk_bytes=256)                      # three pieces through the backend, not the world == 1 short-cut
    ev = D.group_evidence()
    D.enable_weight_broadcast()
    torch.save({"before": before, "grouped": D._grouped(), "ev": ev, "obj": D.broadcast_object({"a": 1}), "blob": ar.blob,
                "enabled": D.weight_broadcast_enabled()}, os.path.join(out, "one.pt"))
    dist.destroy_process_group()


def test_one_rank_group_hook_runs_the_collectives(tmp_path):
    """GDF_RCCL_ONE_RANK=1 (the hook the GPU tests use to put RCCL itself under the N-rank code on a 1-GPU box): with a one-rank group the
    world == 1 short-cuts are NOT taken; without a group the hook changes nothing."""
    mp.spawn(_one_rank_worker, args=(29500 + (os.getpid() + 977) % 2000, str(tmp_path)), nprocs=1, join=True)
    r = torch.load(os.path.join(tmp_path, "one.pt"), weights_only=False)
    assert r["before"] == (False, None)
    assert r["grouped"] and r["enabled"]
    assert r["ev"]["backend"] == "gloo" and r["ev"]["world_size"] == 1 and r["ev"]["ranks_seen"] == [0]
    assert r["obj"] == {"a": 1} and torch.equal(r["blob"], torch.arange(700, dtype=torch.uint8))
