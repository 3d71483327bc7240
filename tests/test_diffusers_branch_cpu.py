"""CPU side of the real-checkpoint boundary and of the N-rank front door (item 2):

  * tests/fake_diffusers (test infrastructure) — its key fixtures are the reference's OWN model classes' names / shapes / configs
    (tests/golden/gen_diffusers_keys.py), and the product's config mappers read them back into the architecture descriptors the
    native models are built from (components/native.py config_from_diffusers, components/models.py flux_ / pixart_config_from_diffusers);
  * the scheduler probes (components/models.py scheduler_noise_scalars / scheduler_step_scalars) against the closed forms of the
    scheduler families the reference configures (feature/components/models.py:26,38,51; diffusion_feature.py:371-380, :477-485);
  * components/dist.py self_launch: no rank outlives the parent (SIGTERM / failure), one retry on a taken rendezvous port, per-rank core sets.
Nothing here needs a GPU or calls into libgdf.
"""
import json
import os
import signal
import subprocess
import sys
import textwrap
import time
import types

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FAKE = os.path.join(ROOT, "tests", "fake_diffusers")
for p in (os.path.join(ROOT, "generic-diffusion-feature_amd"), ROOT):
    if p not in sys.path:
        sys.path.insert(0, p)


@pytest.fixture()
def fake_diffusers(monkeypatch):
    monkeypatch.syspath_prepend(FAKE)
    monkeypatch.setenv("FAKE_DIFFUSERS_CPU_WEIGHTS", "1")
    sys.modules.pop("diffusers", None)
    import diffusers
    assert diffusers.__version__.endswith("+fake")
    diffusers.reset()
    yield diffusers
    sys.modules.pop("diffusers", None)


def test_key_fixtures_are_the_architectures_the_native_models_are_built_for(fake_diffusers):
    from components import models as M
    from components.native import ARCH_CONFIGS, FLUX_CONFIGS, PIXART_CONFIGS, config_from_diffusers
    from oracle import flux_ref as FR, pixart_ref as PR, unet_ref as R, vae_ref as VR
    K = json.load(open(os.path.join(ROOT, "tests", "golden", "diffusers_keys.json")))
    norm = lambda v: tuple(v) if isinstance(v, (list, tuple)) else v
    for tag, ver in (("unet-1-5", "1-5"), ("unet-2-1", "2-1"), ("unet-xl", "xl")):
        cfg = config_from_diffusers(fake_diffusers.FrozenDict(K[tag]["config"]))           # attribute access on a FrozenDict, as diffusers' config gives
        assert {k: norm(v) for k, v in cfg.items()} == {k: norm(v) for k, v in ARCH_CONFIGS[ver].items()}, tag
        # (the int-valued fields of SD1.5's config.json — attention_head_dim = 8, transformer_layers_per_block = 1 — are expanded per level)
        assert {k: tuple(s) for k, s in K[tag]["keys"]} == {k: tuple(s) for k, s in R.param_shapes(R.ARCHS[ver]).items()}
    assert isinstance(K["unet-1-5"]["config"]["attention_head_dim"], int) and K["unet-1-5"]["config"]["transformer_layers_per_block"] == 1
    fc = M.flux_config_from_diffusers(fake_diffusers.FrozenDict(K["flux"]["config"]))
    assert {k: norm(v) for k, v in fc.items()} == {k: norm(v) for k, v in FLUX_CONFIGS["flux"].items()}
    assert {k: tuple(s) for k, s in K["flux"]["keys"]} == {k: tuple(s) for k, s in FR.param_shapes(FR.ARCH_FLUX_DEV).items()}
    pc = M.pixart_config_from_diffusers(fake_diffusers.FrozenDict(K["pixart-sigma"]["config"]))
    assert pc == PIXART_CONFIGS["pixart-sigma"]
    assert M.pixart_config_from_diffusers(fake_diffusers.FrozenDict(K["pixart-sigma"]["config"], sample_size=64, interpolation_scale=None)) == \
        PIXART_CONFIGS["pixart-sigma-512"]
    assert {k: tuple(s) for k, s in K["pixart-sigma"]["keys"]} == {k: tuple(s) for k, s in PR.param_shapes(PR.ARCH_PIXART_SIGMA).items()}
    # AutoencoderKL (un-vendored): the fake's restated module tree against the oracle's, two independent restatements of the published names
    vk = {k: tuple(s) for k, s in fake_diffusers.autoencoder_kl_keys()}
    want = dict(VR.param_shapes(VR.ARCH_SD_VAE)); want.update(VR.dec_param_shapes(VR.ARCH_SD_VAE))
    assert vk == {k: tuple(s) for k, s in want.items()}


def _closed_forms(n=1000):
    betas = torch.linspace(0.00085 ** 0.5, 0.012 ** 0.5, n, dtype=torch.float32) ** 2
    ac = torch.cumprod(1 - betas, 0).double()
    return ac, ((1 - ac) / ac) ** 0.5


def test_noise_and_step_scalars_of_every_scheduler_family(fake_diffusers):
    """prepare_latents' add_noise and the 'vae-out' scheduler.step as (a, b) / (c_sample, c_eps): the probes against the closed forms, driven exactly
    as FeatureExtractor.extract drives the scheduler (set_timesteps(1000) -> get_timesteps(1000, t/1000) -> scale_model_input -> step)."""
    from components import models as M
    from oracle import vae_ref as VR
    D = fake_diffusers
    ac, sig = _closed_forms()
    sd15 = D.StableDiffusionImg2ImgPipeline.from_pretrained("stable-diffusion-v1-5/stable-diffusion-v1-5", torch_dtype=torch.float16, unet=None)
    for t_req in (100, 261, 900):
        # ---- PNDM ('1-5') ----
        sch = sd15.scheduler
        sch.set_timesteps(1000, device="cpu")
        sd15.scheduler = sch
        ts, _ = sd15.get_timesteps(1000, t_req / 1000, "cpu")
        t = ts[:1]
        assert int(t) == t_req + 1                                   # leading spacing with steps_offset 1: "t = 100" is timestep 101
        a, b = M.scheduler_noise_scalars(sch, t.repeat(3))
        assert abs(a - float(ac[int(t)] ** 0.5)) < 1e-6 and abs(b - float((1 - ac[int(t)]) ** 0.5)) < 1e-6
        cs, ce = M.scheduler_step_scalars(sch, t)
        ws, we = VR.pndm_first_step_scalars(ac, int(t), int(t) - 1)
        assert abs(cs - ws) < 1e-6 and abs(ce - we) < 1e-6
        x, e = torch.randn(2, 4, 8, 8, dtype=torch.float64), torch.randn(2, 4, 8, 8, dtype=torch.float64)
        import copy
        assert torch.allclose(copy.deepcopy(sch).step(e, t, x, return_dict=False)[0], cs * x + ce * e, atol=1e-9)
        # ---- EulerDiscrete, made as the product makes it for '2-1' / 'xl': from_config of the pipeline's scheduler config ----
        eu = D.EulerDiscreteScheduler.from_config(sd15.scheduler.config)
        assert eu.config.timestep_spacing == "leading" and eu.config.steps_offset == 1 and eu.config.beta_schedule == "scaled_linear"
        holder = types.SimpleNamespace(scheduler=eu)
        eu.set_timesteps(1000, device="cpu")
        ts, _ = D.StableDiffusionXLImg2ImgPipeline.get_timesteps(holder, 1000, t_req / 1000, "cpu")
        t = ts[:1]
        assert float(t) == t_req and eu.begin_index == 1000 - t_req   # Euler's list is [1000 .. 1] (PNDM's repeats 999, hence its + 1)
        a, b = M.scheduler_noise_scalars(eu, t.repeat(2))
        s_t = float(sig[t_req])
        assert a == 1.0 and abs(b - s_t) < 1e-5 * s_t
        x = torch.randn(1, 4, 4, 4)
        assert torch.allclose(eu.scale_model_input(x, t), x / (s_t ** 2 + 1) ** 0.5, rtol=1e-5)
        cs, ce = M.scheduler_step_scalars(eu, t)
        ws, we = VR.euler_step_scalars(s_t, float(sig[t_req - 1]))
        assert abs(cs - ws) < 1e-6 and abs(ce - we) < 1e-4 * abs(we)   # (the scheduler keeps float32 sigmas)
        # ---- DPMSolverMultistep (the PixArt pipelines): alpha_t x + sigma_t noise — NOT the Euler rule, although the class has `sigmas` too ----
        dp = D.DPMSolverMultistepScheduler(beta_start=0.0001, beta_end=0.02, beta_schedule="linear")
        dp.set_timesteps(1000, device="cpu")
        holder = types.SimpleNamespace(scheduler=dp)
        ts, _ = M._img2img_get_timesteps(holder, 1000, t_req / 1000, "cpu")
        t = ts[:1]
        acl = torch.cumprod(1 - torch.linspace(0.0001, 0.02, 1000, dtype=torch.float32), 0).double()
        a, b = M.scheduler_noise_scalars(dp, t.repeat(2))
        assert abs(a - float(acl[int(t)] ** 0.5)) < 1e-5 and abs(b - float((1 - acl[int(t)]) ** 0.5)) < 1e-5 and abs(a * a + b * b - 1) < 1e-5
    # ---- v-prediction configurations (stabilityai/stable-diffusion-2-1 at 768^2 ships one): the step is still linear in (sample, output) and the
    #      probes find the other pair of coefficients; closed forms from the published update rules ----
    t_req = 261
    pv = D.PNDMScheduler.from_config(dict(sd15.scheduler.config, prediction_type="v_prediction"))
    pv.set_timesteps(1000, device="cpu")
    holder = types.SimpleNamespace(scheduler=pv)
    ts, _ = D.StableDiffusionImg2ImgPipeline.get_timesteps(holder, 1000, t_req / 1000, "cpu")
    t = ts[:1]
    a_t, a_p = float(ac[int(t)]), float(ac[int(t) - 1])
    ws, we = VR.pndm_first_step_scalars(ac, int(t), int(t) - 1)          # epsilon form: prev = ws x + we eps, with eps = sqrt(a_t) v + sqrt(1 - a_t) x
    cs, ce = M.scheduler_step_scalars(pv, t)
    assert abs(cs - (ws + we * (1 - a_t) ** 0.5)) < 1e-6 and abs(ce - we * a_t ** 0.5) < 1e-6
    assert abs(ce - M.scheduler_step_scalars(sd15.scheduler, t)[1]) > 0.1 * abs(ce)   # (and it is NOT the epsilon pair)
    ev = D.EulerDiscreteScheduler.from_config(dict(sd15.scheduler.config, prediction_type="v_prediction"))
    ev.set_timesteps(1000, device="cpu")
    holder = types.SimpleNamespace(scheduler=ev)
    ts, _ = D.StableDiffusionXLImg2ImgPipeline.get_timesteps(holder, 1000, t_req / 1000, "cpu")
    t = ts[:1]
    s_t, s_n = float(sig[t_req]), float(sig[t_req - 1])
    cs, ce = M.scheduler_step_scalars(ev, t)
    assert abs(cs - (1 + s_t / (s_t ** 2 + 1) * (s_n - s_t))) < 1e-5 and abs(ce - (s_n - s_t) / (s_t ** 2 + 1) ** 0.5) < 1e-5
    # a step that is NOT linear (clipping / thresholding of the predicted sample) is refused, not mis-modelled
    class Clipping:
        def step(self, e, t, x, return_dict=False):
            return ((x - e).clamp(-1, 1),)
    with pytest.raises(NotImplementedError):
        M.scheduler_step_scalars(Clipping(), torch.tensor([5]))
    # a scheduler whose add_noise is not linear is refused rather than silently mis-modelled
    class Bad:
        def add_noise(self, x, n, t):
            return x * x + n
    with pytest.raises(NotImplementedError):
        M.scheduler_noise_scalars(Bad(), torch.tensor([5]))


def test_fake_pipelines_have_the_stock_surface(fake_diffusers):
    D = fake_diffusers
    p = D.PixArtSigmaPipeline.from_pretrained("PixArt-alpha/PixArt-Sigma-XL-2-1024-MS", torch_dtype=torch.float16)
    assert not hasattr(p, "get_timesteps") and not hasattr(p, "prepare_latents")          # text-to-image pipeline: the product supplies both
    sd = p.transformer.state_dict()
    assert "pos_embed.pos_embed" not in sd and sd["scale_shift_table"].shape == (2, 1152) and sd["adaln_single.linear.weight"].dtype == torch.float16
    assert D.CALLS[-1][0] == "PixArtSigmaPipeline.from_pretrained" and D.PIPES[-1] is p and p.original["transformer"] is p.transformer
    from PIL import Image
    import numpy as np
    im = Image.fromarray((np.random.RandomState(0).rand(40, 56, 3) * 255).astype(np.uint8))
    x = D.VaeImageProcessor().preprocess(im)
    assert tuple(x.shape) == (1, 3, 40, 56) and x.dtype == torch.float32 and -1.0 <= float(x.min()) < -0.9 and 0.9 < float(x.max()) <= 1.0
    y = D.VaeImageProcessor().preprocess([x[0], x[0]])                                    # tensors already in [-1, 1]: not normalised twice
    assert tuple(y.shape) == (2, 3, 40, 56) and torch.equal(y[0], x[0])


def test_versions_and_dtypes_outside_the_native_path_are_refused_like_the_reference(monkeypatch):
    """reference models.py:11-16 (dtype strings), :173-174 (unknown version); `if` / `hunyuan` exist there but are outside the hot path"""
    from components import models as M
    monkeypatch.delenv("GDF_SYNTHETIC_WEIGHTS", raising=False)
    for v in ("if", "hunyuan", "sd3", ""):
        with pytest.raises(NotImplementedError):
            M.get_diffusion_model(v, "float16", device="cpu")
    with pytest.raises(NotImplementedError):
        M.get_diffusion_model("xl", "bfloat16", device="cpu")
    assert M._HF["2-1"][0] == "stabilityai/stable-diffusion-2-1-base" and M._HF["pgv2"] == ("playgroundai/playground-v2-1024px-aesthetic", "StableDiffusionXLImg2ImgPipeline")


# ---- components/dist.py self_launch ------------------------------------------------------------------------------------------------
_RANK_SCRIPT = textwrap.dedent("""
    import os, sys, time
    out = sys.argv[1]; mode = sys.argv[2]
    r = os.environ["RANK"]
    open(os.path.join(out, f"pid{r}.{os.getpid()}"), "w").write(os.environ.get("GDF_RANK_CORES", ""))
    if mode == "bind":
        flag = os.path.join(out, "tried")
        if r == "0" and not os.path.exists(flag):
            open(flag, "w").write(os.environ["MASTER_PORT"]); sys.exit(97)
        if r == "0":
            open(os.path.join(out, "second_port"), "w").write(os.environ["MASTER_PORT"])
        time.sleep(0.3 if os.path.exists(flag) else 30); sys.exit(0)
    if mode == "sleep":
        time.sleep(120)
    if mode == "ok":
        sys.exit(0)
""")


def _alive(pid):
    try:
        os.kill(pid, 0)
    except ProcessLookupError:
        return False
    try:                                               # a zombie still answers kill(0)
        return open(f"/proc/{pid}/stat").read().split(") ")[1][0] != "Z"
    except FileNotFoundError:
        return False


def _pids(d):
    return [int(f.split(".")[1]) for f in os.listdir(d) if f.startswith("pid")]


def test_self_launch_retries_once_on_a_taken_rendezvous_port_and_pins_cores(tmp_path):
    from components import dist as D
    script = tmp_path / "rank.py"
    script.write_text(_RANK_SCRIPT)
    rc = D.self_launch(str(script), [str(tmp_path), "bind"], 2)
    assert rc == 0
    assert (tmp_path / "tried").read_text() != (tmp_path / "second_port").read_text()       # second attempt on ANOTHER port
    assert len(_pids(tmp_path)) == 4 and not any(_alive(p) for p in _pids(tmp_path))          # two attempts x two ranks, none left
    cores = sorted(os.sched_getaffinity(0))
    sets = D._rank_core_sets(2)
    if len(cores) >= 2:
        assert sorted(sets[0] + sets[1]) == cores and not set(sets[0]) & set(sets[1])
        got = {f.split(".")[0]: (tmp_path / f).read_text() for f in os.listdir(tmp_path) if f.startswith("pid")}
        assert got["pid0"] == ",".join(map(str, sets[0])) and got["pid1"] == ",".join(map(str, sets[1]))
        # pin_rank_cores in a child: the process ends up on exactly its share
        code = ("import os,sys; sys.path.insert(0, %r); from components import dist as D; print(D.pin_rank_cores()); "
                "print(sorted(os.sched_getaffinity(0)))" % os.path.join(ROOT, "generic-diffusion-feature_amd"))
        r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, GDF_RANK_CORES=",".join(map(str, sets[1]))), capture_output=True, text=True)
        assert r.returncode == 0 and r.stdout.splitlines() == [str(sets[1]), str(sets[1])], r.stderr[-500:]
        r = subprocess.run([sys.executable, "-c", code], env={k: v for k, v in dict(os.environ, LOCAL_RANK="0", WORLD_SIZE="2", LOCAL_WORLD_SIZE="2").items()
                                                               if k != "GDF_RANK_CORES"}, capture_output=True, text=True)   # torchrun-style environment
        assert r.returncode == 0 and r.stdout.splitlines()[1] == str(sets[0]), r.stderr[-500:]


def test_self_launch_parent_killed_takes_every_rank_down(tmp_path):
    """SIGTERM to the parent (a CI timeout) must not leave N rank processes holding their GPUs."""
    script = tmp_path / "rank.py"
    script.write_text(_RANK_SCRIPT)
    code = ("import sys; sys.path.insert(0, %r); from components import dist as D; sys.exit(D.self_launch(%r, [%r, 'sleep'], 3))"
            % (os.path.join(ROOT, "generic-diffusion-feature_amd"), str(script), str(tmp_path)))
    p = subprocess.Popen([sys.executable, "-c", code], stderr=subprocess.PIPE, text=True)
    t0 = time.time()
    while len(_pids(tmp_path)) < 3 and time.time() - t0 < 60:
        time.sleep(0.05)
    pids = _pids(tmp_path)
    assert len(pids) == 3 and all(_alive(x) for x in pids)
    p.send_signal(signal.SIGTERM)
    err = p.communicate(timeout=60)[1]
    assert p.returncode == 128 + signal.SIGTERM, (p.returncode, err[-500:])
    assert "stopping 3 rank processes" in err
    assert not any(_alive(x) for x in pids)


def test_self_launch_failure_of_one_rank_stops_the_others(tmp_path):
    from components import dist as D
    script = tmp_path / "rank.py"
    script.write_text(_RANK_SCRIPT + textwrap.dedent("""
        if mode == "fail":
            if r == "1":
                sys.exit(5)
            time.sleep(120)
    """))
    t0 = time.time()
    rc = D.self_launch(str(script), [str(tmp_path), "fail"], 3)
    assert rc == 5 and time.time() - t0 < 60
    assert not any(_alive(x) for x in _pids(tmp_path))


def test_offline_lora_without_a_file_name_is_a_local_model_directory():
    """reference feature/components/models.py:21-22 (and the same two lines in every version branch): `offline_lora and not offline_lora_filename`
    -> `model_id = offline_lora`; with a file name the hub id stays and diffusion_feature.py:50-52 loads the LoRA on top."""
    from components import models as M
    assert M._model_id("stabilityai/stable-diffusion-xl-base-1.0", None, None) == "stabilityai/stable-diffusion-xl-base-1.0"
    assert M._model_id("stabilityai/stable-diffusion-xl-base-1.0", "/data/my-sdxl", None) == "/data/my-sdxl"
    assert M._model_id("stabilityai/stable-diffusion-xl-base-1.0", "/data/lora", "pytorch_lora_weights.safetensors") == "stabilityai/stable-diffusion-xl-base-1.0"
    assert M._model_id("x", "", None) == "x"


def test_prompts_longer_than_the_tokenizer_window_are_encoded_in_windows(fake_diffusers):
    """FeatureExtractor.encode_prompt's > 70-word branch (reference feature/diffusion_feature.py:165-171 -> components/encode_long_prompt.py:5-40): the
    prompt is tokenised without truncation, the (shorter) negative prompt padded to the same id count, both rows encoded window by window (77, 77, ...,
    rest) and concatenated along the token axis."""
    import diffusion_feature as DF
    D = fake_diffusers
    pipe = D.StableDiffusionImg2ImgPipeline.from_pretrained("stable-diffusion-v1-5/stable-diffusion-v1-5", torch_dtype=torch.float32, unet=None)
    words = " ".join(f"w{i}" for i in range(170))                      # 170 words + BOS + EOS = 172 ids = 77 + 77 + 18
    D.CALLS.clear()
    pe, ne = DF._chunked_prompt_embeds(pipe, words, "", "cpu")
    assert pe.shape == ne.shape == (1, 172, 768)
    toks = [c[1] for c in D.CALLS if c[0] == "tokenizer"]
    assert toks == [dict(words=170, truncation=False, padding=False, max_length=None), dict(words=0, truncation=False, padding="max_length", max_length=172)]
    assert [c[1]["tokens"] for c in D.CALLS if c[0] == "text_encoder"] == [77, 77, 18, 77, 77, 18]     # the prompt's windows, then the negative prompt's
    ids = pipe.tokenizer(words, return_tensors="pt", truncation=False).input_ids
    want = torch.cat([pipe.text_encoder(ids[:, a:a + 77])[0] for a in (0, 77, 154)], 1)
    assert torch.equal(pe, want)
    assert not torch.allclose(pe[:, 77:154], pipe.text_encoder(ids)[0][:, 77:154])      # one pass over all ids is NOT the same thing (positions restart per window)
    # the negative prompt decides the length when it has more words
    pe2, ne2 = DF._chunked_prompt_embeds(pipe, "a cat", " ".join(["no"] * 90), "cpu")
    assert pe2.shape == ne2.shape == (1, 92, 768)
    # the method: 4-tuple with no pooled embeddings; <= 70 words still go through pipe.encode_prompt
    fx = DF.FeatureExtractor.__new__(DF.FeatureExtractor)
    torch.nn.Module.__init__(fx)
    fx.pipe, fx.device, fx.version = pipe, "cpu", "1-5"
    out = fx.encode_prompt(words)
    assert torch.equal(out[0], pe) and out[2] is None and out[3] is None
    D.CALLS.clear()
    short = fx.encode_prompt("a photo of a cat")
    assert short[0].shape == (1, 77, 768) and [c[0] for c in D.CALLS] == ["encode_prompt"]
