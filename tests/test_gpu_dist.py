"""Multi-process / multi-thread use of the native path on ONE GPU box (-m gpu).

The driver owns the 8-GPU runs; what can be checked here is that the data-parallel code path is correct by construction:
two ranks (fresh child processes, both on cuda:0 through the GDF_SHARE_GPU / GDF_BENCH_SHARE_GPU test hooks, gloo instead
of RCCL) must reproduce the single-process result bit for bit, with rank 1 receiving its weights ONLY through the broadcast
of rank 0's device arena; and two extractors in two Python threads of one process — the reference's own multi-device mode
(correspondence/correspondence/aggregation_network.py:67-95) — must not disturb each other."""
import json
import os
import socket
import subprocess
import sys
import threading

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _tree(d):
    out = {}
    for root, _, files in os.walk(d):
        for f in files:
            out[os.path.relpath(os.path.join(root, f), d)] = np.load(os.path.join(root, f))
    return out


def test_two_rank_cli_equals_single_process(tmp_path):
    from PIL import Image
    rs = np.random.RandomState(0)
    (tmp_path / "imgs").mkdir()
    for n in "abcd":
        Image.fromarray((rs.rand(96, 120, 3) * 255).astype(np.uint8)).save(tmp_path / "imgs" / f"{n}.png")
    (tmp_path / "prompt.txt").write_text("a photo of a cat")
    (tmp_path / "layers.json").write_text(json.dumps({"up-level1-repeat2-res-out": True, "up-level3-repeat0-vit-block0-self-k": True}))
    base = [os.path.join(ROOT, "extract_feature.py"), "--layer", str(tmp_path / "layers.json"), "--version", "1-5", "--img_size", "256",
            "--t", "100", "-b", "2", "--input_dir", str(tmp_path / "imgs" / "*.png"), "--prompt_file", str(tmp_path / "prompt.txt")]
    env = dict(os.environ, GDF_SYNTHETIC_WEIGHTS="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable] + base + ["--output_dir", str(tmp_path / "one")], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    env2 = dict(env, GDF_SHARE_GPU="1")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", str(_free_port())] + base + ["--output_dir", str(tmp_path / "two")],
                       env=env2, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    one, two = _tree(tmp_path / "one"), _tree(tmp_path / "two")
    assert sorted(one) == sorted(two) and len(one) == 8                      # 2 layers x 4 images, <split><GLOBAL index> names
    for k in one:
        assert one[k].dtype == two[k].dtype and np.array_equal(one[k].view(np.uint16), two[k].view(np.uint16)), k
        assert np.abs(one[k].astype(np.float32)).max() > 0                   # rank 1 really received weights (zeros otherwise)


def test_two_rank_bench_line():
    env = dict(os.environ, GDF_BENCH_SHARE_GPU="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
                        "--version", "1-5", "--batch", "4", "--no-cpu-baseline"], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    line = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["scaling"] == "weak" and line["config"]["global_batch"] == 8 and line["value"] > 0
    assert line["roofline"]["launches"] > 0 and line["roofline"]["achieved"] > 0


def _one_rank_env(**extra):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE")}
    env.update(GDF_RCCL_ONE_RANK="1", RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()),
               HSA_ENABLE_IPC_MODE_LEGACY="0", **extra)
    return env


def test_rccl_one_rank_group_runs_the_n_rank_bench_path():
    """The share-GPU tests above run the N-rank code on gloo (RCCL refuses two ranks on one device).  This one puts the REAL backend under
    the same calls: a one-rank RCCL group (GDF_RCCL_ONE_RANK=1) — init_process_group('nccl', device_id=...), the int64 size all_reduce and
    the 512-MiB-piece broadcasts of the device weight arena, the one-hot all_reduce + all_gather_object of the rank evidence, the barriers
    around the timed region, the float64 MAX all_reduce of the step times, destroy_process_group — on the GPU."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1", "--version", "1-5", "--batch", "4",
                        "--no-cpu-baseline", "--no-extras"], env=_one_rank_env(), capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    # ONE line on stdout, and it is the JSON: RCCL's own version banner (C stdio on stdout, flushed at exit, i.e. after the line) must not reach it
    assert len(r.stdout.splitlines()) == 1, r.stdout[-600:]
    line = json.loads(r.stdout)
    g = line["rccl"]
    assert g["backend"].startswith("rccl") and g["library_version"], g
    assert g["world_size"] == 1 and g["ranks_seen"] == [0] and g["distinct_devices"] == 1, g
    assert line["n_gpus"] == 1 and line["value"] > 0
    assert line["config"]["weights_broadcast_gb_per_s"] is not None and line["config"]["weights_broadcast_gb_per_s"] > 0     # the broadcast ran
    pr = line["config"]["per_rank_ms_per_step"]
    assert pr["min"] == pr["max"] > 0                                                                                   # MAX all_reduce of (dt, -dt) over one rank


def test_rccl_one_rank_group_cli_same_files(tmp_path):
    """extract_feature.py through the N-rank branch (RCCL group of one: config broadcast_object, weight-arena broadcast, shard_range) writes
    the files of the plain single-process run bit for bit."""
    from PIL import Image
    rs = np.random.RandomState(3)
    (tmp_path / "imgs").mkdir()
    for n in "abc":
        Image.fromarray((rs.rand(80, 96, 3) * 255).astype(np.uint8)).save(tmp_path / "imgs" / f"{n}.png")
    (tmp_path / "prompt.txt").write_text("a photo of a dog")
    (tmp_path / "layers.json").write_text(json.dumps({"up-level1-repeat2-res-out": True, "up-level3-repeat0-vit-block0-self-k": True}))
    base = [sys.executable, os.path.join(ROOT, "extract_feature.py"), "--layer", str(tmp_path / "layers.json"), "--version", "1-5", "--img_size", "256",
            "--t", "100", "-b", "2", "--input_dir", str(tmp_path / "imgs" / "*.png"), "--prompt_file", str(tmp_path / "prompt.txt")]
    env = {k: v for k, v in dict(os.environ, GDF_SYNTHETIC_WEIGHTS="1", HSA_ENABLE_IPC_MODE_LEGACY="0").items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE")}
    r = subprocess.run(base + ["--output_dir", str(tmp_path / "plain")], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    r = subprocess.run(base + ["--output_dir", str(tmp_path / "rccl")], env=_one_rank_env(GDF_SYNTHETIC_WEIGHTS="1"), capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    a, b = _tree(tmp_path / "plain"), _tree(tmp_path / "rccl")
    assert sorted(a) == sorted(b) and len(a) == 6
    for k in a:
        assert np.array_equal(a[k].view(np.uint16), b[k].view(np.uint16)), k


def test_two_threads_two_extractors(monkeypatch):
    """One extractor per Python thread (reference aggregation_network.py:67-95), here both on cuda:0: concurrent plan
    creation, first-launch attribute setup, graph capture and replay must give each thread exactly its sequential result."""
    monkeypatch.setenv("GDF_SYNTHETIC_WEIGHTS", "1")
    import diffusion_feature
    layer = {"up-level1-repeat1-vit-block0-cross-q": True, "up-level2-repeat2-res-out": True}
    lats = [torch.randn(2, 4, 32, 32, generator=torch.Generator().manual_seed(s)).half() for s in (0, 1)]

    def make(seed):
        monkeypatch.setenv("GDF_SYNTHETIC_SEED", str(seed))
        return diffusion_feature.FeatureExtractor(layer=layer, version='1-5', img_size=256, device='cuda:0')
    want = []
    for s in (0, 1):                                                         # sequential reference, fresh extractors
        df = make(s)
        f = df.extract(df.encode_prompt('a photo of a cat'), batch_size=2, image=lats[s], image_type='latents', t=100)
        want.append({k: v.clone() for k, v in f.items()})
        del df, f
    dfs = [make(0), make(1)]
    got, errs = [None, None], []
    gate = threading.Barrier(2)

    def work(i):
        try:
            df = dfs[i]
            prompt = df.encode_prompt('a photo of a cat')
            gate.wait()
            for _ in range(4):                                               # eager warm-up, capture, replays — all overlapping
                f = df.extract(prompt, batch_size=2, image=lats[i], image_type='latents', t=100)
            torch.cuda.synchronize()
            got[i] = {k: v.clone() for k, v in f.items()}
        except Exception as e:                                               # surfaced in the main thread
            errs.append(e)
    th = [threading.Thread(target=work, args=(i,)) for i in range(2)]
    [t.start() for t in th]
    [t.join() for t in th]
    assert not errs, errs
    for i in range(2):
        for k in want[i]:
            assert torch.equal(got[i][k], want[i][k]), (i, k)
    assert not torch.equal(want[0]["up-level2-repeat2-res-out"], want[1]["up-level2-repeat2-res-out"])


def test_four_rank_cli_with_empty_shard_and_vae_out(tmp_path):
    """Row (e) launch-readiness: FOUR ranks on one GPU through extract_feature.py with only THREE images — one
    rank's shard is empty — and 'vae-out' in the layer config: the decoder's weight fill is a collective, so it must be built on every rank
    in the constructor (a rank that never calls extract() used to leave the others blocked in the broadcast).  Output = the single-process
    files, bit for bit."""
    from PIL import Image
    rs = np.random.RandomState(1)
    (tmp_path / "imgs").mkdir()
    for n in "abc":
        Image.fromarray((rs.rand(64, 64, 3) * 255).astype(np.uint8)).save(tmp_path / "imgs" / f"{n}.png")
    (tmp_path / "prompt.txt").write_text("a photo of a dog")
    (tmp_path / "layers.json").write_text(json.dumps({"up-level1-repeat2-res-out": True, "vae-out": True}))
    # (-b 1: every forward is a batch of one in both runs — plans of different batch sizes pick different tiles / split-K factors and agree
    #  to fp32 summation order only, not bit for bit)
    base = [os.path.join(ROOT, "extract_feature.py"), "--layer", str(tmp_path / "layers.json"), "--version", "1-5", "--img_size", "128",
            "--t", "100", "-b", "1", "--input_dir", str(tmp_path / "imgs" / "*.png"), "--prompt_file", str(tmp_path / "prompt.txt")]
    env = dict(os.environ, GDF_SYNTHETIC_WEIGHTS="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable] + base + ["--output_dir", str(tmp_path / "one")], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    # plain `python3 extract_feature.py --gpus 4 ...` (no torchrun): the CLI starts its four ranks itself (components/dist.py self_launch)
    env4 = {k: v for k, v in dict(env, GDF_SHARE_GPU="1").items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    r = subprocess.run([sys.executable] + base + ["--gpus", "4", "--output_dir", str(tmp_path / "four")],
                       env=env4, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    one, four = _tree(tmp_path / "one"), _tree(tmp_path / "four")
    assert sorted(one) == sorted(four) and len(one) == 6                     # 2 layers x 3 images
    for k in one:
        assert np.array_equal(one[k].view(np.uint16), four[k].view(np.uint16)), k


def test_bench_front_door_failure_is_nonzero():
    """a rank that dies after the rendezvous (the other one is waiting in a collective) brings the whole self-launched job down: non-zero
    exit code, no JSON line, no orphaned rank"""
    env = {k: v for k, v in dict(os.environ, GDF_BENCH_SHARE_GPU="1", GDF_TEST_HOOKS="1", GDF_TEST_FAIL_RANK="1").items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--version", "1-5", "--batch", "2",
                        "--no-cpu-baseline"], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 3 and not [l for l in r.stdout.splitlines() if l.startswith("{")], (r.returncode, r.stderr[-1500:])
    assert "rank 1 exited with code 3" in r.stderr


def test_four_rank_bench_line():
    """bench.py --gpus 4 on one GPU (GDF_BENCH_SHARE_GPU=1): the line carries the broadcast time and the per-rank step times, and the whole-job
    value is 4 ranks x batch x steps / the slowest rank's time.  Round 6: the N > 1 line is COMPLETE — `roofline` and
    `cpu_baseline` (timed by rank 0 after the process group is gone) are in it, and every rank was pinned to its own share of the host cores."""
    env = dict(os.environ, GDF_BENCH_SHARE_GPU="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    env = {k: v for k, v in env.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    # the command shape the driver uses: PLAIN python3 bench.py --gpus N (no torchrun) — bench.py starts its four ranks itself
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4", "--steps", "2", "--warmup", "1",
                        "--version", "1-5", "--batch", "2"], env=env, capture_output=True, text=True, timeout=1200)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1                                                   # ONE JSON line on stdout: rank 0's
    line = json.loads(lines[0])
    rf, cb = line["roofline"], line["cpu_baseline"]
    assert rf["bound"] == "mfma" and rf["achieved"] > 0 and rf["peak"] == 2500.0 and 0 < rf["frac"] < 1 and rf["launches"] > 0
    assert cb["value"] > 0 and cb["kind"] == "port" and cb["cores"] >= 1 and "after destroy_process_group" in cb["host"]["when"]
    ncpu = len(os.sched_getaffinity(0))
    if ncpu >= 4:                                                            # every rank on its own quarter of the cores this test may use
        assert cb["host"]["rank0_affinity_cpus"] == ncpu // 4 + (1 if ncpu % 4 else 0) and cb["cores"] <= cb["host"]["rank0_affinity_cpus"]
        assert line["config"]["rank_cpu_affinity"]["rank0_cpus"] == cb["host"]["rank0_affinity_cpus"]
    c = line["config"]
    g = line["rccl"]
    assert g["world_size"] == 4 and g["ranks_seen"] == [0, 1, 2, 3] and g["backend"] == "gloo"      # (share-GPU test hook: gloo; RCCL on a real node)
    assert c["launched_by"].startswith("bench.py itself") and c["max_inflight_forwards"] == 4 and c["weights_broadcast_gb_per_s"] > 0
    assert line["n_gpus"] == 4 and line["scaling"] == "weak" and c["global_batch"] == 8 and line["value"] > 0
    assert c["weights_broadcast_s"] >= 0 and 0 < c["per_rank_ms_per_step"]["min"] <= c["per_rank_ms_per_step"]["max"]
    assert abs(line["value"] - 8 * 1e3 / c["per_rank_ms_per_step"]["max"]) < 0.02 * line["value"]
    assert "plans" not in line and "e2e" not in line                         # the extra legs are single-GPU only


def test_four_rank_cli_uneven_shards_bit_equal_to_one_process(tmp_path):
    """MORE images than ranks, uneven shards.  SEVEN images over FOUR ranks (shards of 2, 2, 2, 1), batch size 2; the last
    rank runs a batch-1 plan, as the single process does for its ragged last batch: every file must equal the single-process file bit for
    bit.  (The VAE sampling / add_noise draws are per batch position, like the reference's `prepare_latents`, so the CLI cannot move an image
    to another batch index without changing its latents; that the DENOISER gives identical bits at every batch index is asserted on
    identical latents in test_per_sample_bits_do_not_depend_on_batch_index below and at SDXL B = 16 in test_gpu_fullsize.py.)"""
    from PIL import Image
    rs = np.random.RandomState(2)
    (tmp_path / "imgs").mkdir()
    names = "abcdefg"
    for n in names:
        Image.fromarray((rs.rand(64, 64, 3) * 255).astype(np.uint8)).save(tmp_path / "imgs" / f"{n}.png")
    (tmp_path / "prompt.txt").write_text("a photo of a bird")
    layers = {"up-level1-repeat2-res-out": True, "up-level2-repeat1-vit-block0-cross-q": True, "down-level0-repeat0-res-increment": True}
    (tmp_path / "layers.json").write_text(json.dumps(layers))
    base = [os.path.join(ROOT, "extract_feature.py"), "--layer", str(tmp_path / "layers.json"), "--version", "1-5", "--img_size", "128",
            "--t", "100", "-b", "2", "--input_dir", str(tmp_path / "imgs" / "*.png"), "--prompt_file", str(tmp_path / "prompt.txt")]
    env = {k: v for k, v in dict(os.environ, GDF_SYNTHETIC_WEIGHTS="1", HSA_ENABLE_IPC_MODE_LEGACY="0").items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    r = subprocess.run([sys.executable] + base + ["--output_dir", str(tmp_path / "one")], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    r = subprocess.run([sys.executable] + base + ["--gpus", "4", "--output_dir", str(tmp_path / "four")], env=dict(env, GDF_SHARE_GPU="1"),
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    one, four = _tree(tmp_path / "one"), _tree(tmp_path / "four")
    assert sorted(one) == sorted(four) and len(one) == 3 * 7
    # shards: rank 0 = images 0-1, rank 1 = 2-3, rank 2 = 4-5 (batch-2 plans, like the single process's batches 0-1, 2-3, 4-5), rank 3 = image 6
    # (a batch-1 plan; the single process ALSO runs image 6 as its last, ragged batch of one): every file must be bit-identical
    for k in one:
        assert np.array_equal(one[k].view(np.uint16), four[k].view(np.uint16)), k


def test_per_sample_bits_do_not_depend_on_batch_index(monkeypatch):
    """The same pre-noised latent at batch index 0, 1, 2 of a batch of three different latents (and alone in a batch whose other members
    change) gives bit-identical features: the plan's kernels are per-sample position independent (round 5: the stacked time-embedding linear
    used to round rows {0, 3} and {1, 2} of a batch differently — tools/op_batch_position.py)."""
    monkeypatch.setenv("GDF_SYNTHETIC_WEIGHTS", "1")
    import diffusion_feature
    layer = {"down-level0-repeat0-res-increment": True, "up-level1-repeat1-vit-block0-cross-q": True, "up-level2-repeat2-res-out": True,
             "mid-vit-block0-ffn-inner": True}
    df = diffusion_feature.FeatureExtractor(layer=layer, version='1-5', img_size=256, device='cuda:0')
    prompt = df.encode_prompt('a photo of a cat')
    lat = [torch.randn(4, 32, 32, generator=torch.Generator().manual_seed(s)).half() for s in (0, 1, 2, 3)]
    runs = {}
    for order in ((0, 1, 2, 3), (3, 0, 1, 2), (2, 3, 0, 1), (1, 2, 3, 0)):
        x = torch.stack([lat[i] for i in order])
        f = df.extract(prompt, batch_size=4, image=x, image_type='latents', t=100)
        torch.cuda.synchronize()
        runs[order] = {k: v.clone() for k, v in f.items()}
    base = runs[(0, 1, 2, 3)]
    for order, f in runs.items():
        for pos, i in enumerate(order):
            for k in layer:
                assert torch.equal(f[k][pos], base[k][i]), (order, pos, k)


_FREE_DURING_CAPTURE = r'''
import os, sys, threading, json
ROOT = sys.argv[1]
for p in (ROOT, os.path.join(ROOT, "generic-diffusion-feature_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import torch
from oracle import unet_ref as R
from helpers import cfg_from_oracle_arch
from components.native import NativeUNet, NativeVAEEncoder
arch = R.tiny_arch("xl")
P = {k: v.half() for k, v in R.synth_params(arch, seed=0).items()}
I = R.synth_inputs(arch, 2, 16, seed=1)
u = NativeUNet(cfg_from_oracle_arch(arch), device="cuda:0", precise=False)
u.load_state_dict(P)
ids_all = [i for i in u.hook_names() if not i.endswith("-map")]
g = lambda k: I[k].cuda() if k in I else None
ref = u.forward_raw(g("sample"), g("timestep"), g("ctx"), g("text_embeds"), g("time_ids"), hook_ids=ids_all[:3])[1]
ref = {k: v.clone() for k, v in ref.items()}
stop, errs, stats = threading.Event(), [], dict(captures=0, failures=0, frees=0, mismatches=0)

def capturer():                      # a new hook set = a new plan = eager warm-up, capture, replay — 40 captures back to back
    try:
        for n in range(40):
            ids = ids_all[:3] + [ids_all[3 + n]]
            for _ in range(3):
                h = u.forward_raw(g("sample"), g("timestep"), g("ctx"), g("text_embeds"), g("time_ids"), hook_ids=ids)[1]
            torch.cuda.synchronize()
            stats["mismatches"] += sum(int(not torch.equal(h[k], ref[k])) for k in ref)
            for p in u._plans.values():
                c, l, f = p.graph_stats()
            plan = list(u._plans.values())[-1]
            c, l, f = plan.graph_stats()
            stats["captures"] += c; stats["failures"] += f
            del h
    except Exception as e:
        errs.append(repr(e)[:300])
    finally:
        stop.set()

def freer():                         # models created and destroyed (hipMalloc + hipFree of their weight arenas) as fast as possible
    cfg = dict(in_channels=3, latent_channels=4, block_out_channels=(64, 128, 128), layers_per_block=2, use_quant_conv=1)
    try:
        while not stop.is_set():
            m = NativeVAEEncoder(cfg, device="cuda:0")
            del m
            stats["frees"] += 1
    except Exception as e:
        errs.append(repr(e)[:300])

th = [threading.Thread(target=capturer), threading.Thread(target=freer)]
[t.start() for t in th]; [t.join() for t in th]
# the stream must still be usable afterwards
h = u.forward_raw(g("sample"), g("timestep"), g("ctx"), g("text_embeds"), g("time_ids"), hook_ids=ids_all[:3])[1]
torch.cuda.synchronize()
stats["mismatches"] += sum(int(not torch.equal(h[k], ref[k])) for k in ref)
print(json.dumps(dict(stats, errors=errs)))
'''


def test_model_frees_in_one_thread_do_not_invalidate_captures_in_another(tmp_path):
    """Round 6 (found by the full suite, not by a single test): a native model destroyed in one host thread — hipFree of its weight arena, e.g. a
    pipeline garbage-collected late — while another thread's plan is capturing its hipGraph invalidated that capture on this runtime, and the
    invalidated stream (one of torch's 32 pooled streams) then failed every later plan that was handed it.  The library first kept its own
    allocations / frees out of capture windows (csrc/model.h) and, later in round 6, stopped capturing altogether: graphs are built node by node
    (csrc/launch.h), so there is no capture to invalidate.  Forty graph constructions in one thread against a
    thread that creates and destroys models in a loop: no capture may fail, results bit-equal, the stream usable afterwards.  (GDF_TEST_UNGUARDED=1
    also runs the script with GDF_CAPTURE_GUARD=0: measured in round 6, that process segfaults.)"""
    script = tmp_path / "free_during_capture.py"
    script.write_text(_FREE_DURING_CAPTURE)
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, str(script), ROOT], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    d = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    print("\n[capture guard ON ]", d)
    assert d["errors"] == [] and d["failures"] == 0 and d["mismatches"] == 0 and d["captures"] >= 40 and d["frees"] >= 20, d
    if os.environ.get("GDF_TEST_UNGUARDED") == "1":
        # for the record only (measured in round 6: the unguarded process dies with SIGSEGV inside the runtime): not part of the default run — a
        # crashing GPU process is not something a test suite should do to a shared box
        r0 = subprocess.run([sys.executable, str(script), ROOT], env=dict(env, GDF_CAPTURE_GUARD="0"), capture_output=True, text=True, timeout=900)
        tail = [l for l in r0.stdout.splitlines() if l.startswith("{")]
        print("[capture guard OFF]", tail[-1] if tail else (r0.returncode, r0.stderr[-300:]))


def _json_tail(stdout):
    return json.loads([l for l in stdout.splitlines() if l.startswith("{")][-1])


def test_two_host_threads_two_streams_soak():
    """One extractor per host thread is a supported mode (reference correspondence/correspondence/aggregation_network.py:67-95).  tools/soak.py for
    half a minute from two threads: random versions / sizes / layer sets / batch sizes / plans, every extract repeated (same bits), a fixed probe
    configuration re-run throughout (the bits of the first run), no plan falling back from graph replay, memory returned.  Round 6 found three
    things this way, each fixed at its root: a slot-reuse race in the 256x256 two-group GEMM main loop that only another stream's waves on the CU
    exposed (profiles/r06_concurrent_streams.txt), stream captures invalidated by another thread's device-wide calls (graphs are now BUILT with
    the graph API, csrc/launch.h), and freed plan buffers stranded in the cache of the plan's dead stream."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "soak.py"), "--minutes", "0.5", "--threads", "2"], env=env, capture_output=True, text=True,
                       timeout=900)
    assert r.stdout.strip(), r.stderr[-3000:]
    d = _json_tail(r.stdout)
    print("\n[soak]", {k: d[k] for k in ("lifetimes", "extracts", "graph_captures", "graph_failures", "not_returned_mb")})
    assert r.returncode == 0 and d["ok"], json.dumps(d)[:3000] + r.stderr[-1500:]
    assert d["lifetimes"] >= 40 and d["graph_captures"] >= 40 and d["graph_failures"] == 0 and not d["mismatches"] and not d["probe_drift"] and not d["errors"], d


_SYNC_BESIDE_GRAPH_BUILD = r'''
import json, os, sys, threading, time
ROOT = sys.argv[1]
sys.path.insert(0, os.path.join(ROOT, "generic-diffusion-feature_amd"))
import torch
from components.native import ARCH_CONFIGS, NativeUNet
dev = torch.device("cuda:0"); torch.cuda.set_device(dev)
u = NativeUNet(ARCH_CONFIGS["1-5"], device=dev); u.init_synthetic(seed=0)
ids = [h for h in u.hook_names() if not h.endswith("-map")][::7]
g = torch.Generator(device=dev).manual_seed(3)
ctx = torch.randn(1, 77, 768, generator=g, device=dev).half()
stop = threading.Event(); errs = []; stats = dict(syncs=0, builds=0, failures=0, mismatches=0, allocs=0)

def other():                       # what any other host thread of the process may do at any time
    torch.cuda.set_device(dev)
    try:
        while not stop.is_set():
            torch.cuda.synchronize(); stats["syncs"] += 1                     # hipDeviceSynchronize: illegal beside a stream capture
            t = torch.empty(64 << 20, dtype=torch.uint8, device=dev); del t    # hipMalloc / hipFree traffic
            torch.cuda.empty_cache(); stats["allocs"] += 1
    except Exception as e:
        errs.append("other thread: " + repr(e)[:300])

th = threading.Thread(target=other); th.start()
try:
    for B in (1, 2, 3, 4, 5, 6):      # six plans, each built into a hipGraph on its second forward, hook sets rotating
        x = torch.randn(B, 4, 32, 32, generator=g, device=dev).half()
        c = ctx.expand(B, -1, -1).contiguous()
        ref = None
        for it in range(6):
            n, h = u.forward_raw(x, 100.0, c, hook_ids=ids, shared_ctx=True)
            torch.cuda.current_stream().synchronize()
            cur = {k: v.clone() for k, v in h.items()}
            if ref is None:
                ref = cur
            stats["mismatches"] += sum(int(not torch.equal(cur[k], ref[k])) for k in ref)
            del n, h
        for pl in u._plans.values():
            cap, lau, fail = pl.graph_stats()
        stats["builds"] = sum(pl.graph_stats()[0] for pl in u._plans.values()); stats["failures"] = sum(pl.graph_stats()[2] for pl in u._plans.values())
except Exception as e:
    errs.append("forward thread: " + repr(e)[:300])
stop.set(); th.join()
print(json.dumps(dict(stats, errors=errs)))
'''


def test_device_synchronize_in_another_thread_beside_graph_construction(tmp_path):
    """While one thread's plan is turned into a hipGraph, another thread calls torch.cuda.synchronize(), allocates, frees and empties torch's cache
    in a loop.  Under stream capture (rounds 2-5) the hipDeviceSynchronize raised hipErrorStreamCaptureUnsupported in THAT thread and invalidated
    the capture in the other (tools/micro/thread_race.py); graphs are now built node by node (csrc/launch.h), so both threads run undisturbed:
    no error on either side, every plan replays a graph, bits equal across replays."""
    script = tmp_path / "sync_beside_build.py"
    script.write_text(_SYNC_BESIDE_GRAPH_BUILD)
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, str(script), ROOT], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    d = _json_tail(r.stdout)
    print("\n[sync beside graph build]", d)
    assert d["errors"] == [] and d["failures"] == 0 and d["mismatches"] == 0 and d["builds"] >= 6 and d["syncs"] >= 20, d
