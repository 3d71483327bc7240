"""components/plan_levels.py on the CPU (no GPU, no libgdf): the verify ladder's arithmetic and bookkeeping with a synthetic `run`, and the
chooser's masks.  Reference contract being protected: FeatureStore hands out what the model computed
(/root/reference/feature/components/feature_extractor.py:31-76) — here: within 1e-3 of it."""
import os
import sys
import warnings

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "generic-diffusion-feature_amd"))
from components import plan_levels as PL  # noqa: E402
from components.native import ARCH_CONFIGS  # noqa: E402  (pure data; importing native does not load libgdf)


def _runner(err_by_mask, ids, fail=()):
    """run(mask) -> (noise, hooks) whose relative L2 distance to the full split's result is err_by_mask[mask]"""
    g = torch.Generator().manual_seed(0)
    base = {k: torch.randn(4, 8, 6, 6, generator=g) for k in ids}
    noise = torch.randn(4, 4, 6, 6, generator=g)
    dirs = {k: torch.randn(4, 8, 6, 6, generator=g) for k in ids}
    calls = []

    def run(mask):
        calls.append(mask)
        if mask in fail:
            raise RuntimeError("plan does not exist at this size")
        e = 0.0 if mask == PL.SPLIT_ALL else err_by_mask[mask]
        hooks = {k: base[k] + dirs[k] * (e * base[k].norm() / dirs[k].norm()) for k in ids}
        return noise + (e * noise.norm() / noise.norm()) * 0 + e * noise, hooks
    return run, calls


def test_acceptance_is_arithmetic_per_family():
    xl, sd = PL.VerifyLadder(ARCH_CONFIGS["xl"]), PL.VerifyLadder(ARCH_CONFIGS["1-5"])
    assert abs(xl.accept_bound() - (0.97e-3 ** 2 - 2.7e-4 ** 2) ** 0.5) < 1e-12 and abs(sd.accept_bound() - (0.97e-3 ** 2 - 4.4e-4 ** 2) ** 0.5) < 1e-12
    assert xl.accept_bound() > sd.accept_bound()                       # the family whose full split is further from fp32 gets the tighter bound on d
    lv = xl.levels()
    assert lv[0] == 0 and lv[-1] == PL.SPLIT_ALL and all((a & b) == a for a, b in zip(lv, lv[1:]))     # nested, cheapest first
    assert len(lv) == 8 and lv[-2] == PL.SPLIT_ALL & ~PL.SPLIT_CLASSES["res"]
    assert len(PL.VerifyLadder(ARCH_CONFIGS["1-5"]).levels()) == 8


def test_ladder_keeps_the_first_level_inside_the_bound_and_escalates_from_then_on():
    cfg = ARCH_CONFIGS["xl"]
    L = PL.VerifyLadder(cfg)
    ids = ["mid-vit-block3-ffn-inner", "up-level1-repeat2-res-out"]
    lv = L.levels()
    cur = lv[2]                                                         # the table's choice for an ffn-inner hook: the selective preset
    assert PL.choose_split(cfg, ids) == cur
    errs = {lv[2]: 1.10e-3, lv[3]: 9.4e-4, lv[4]: 9.2e-4, lv[5]: 5e-4, lv[6]: 3e-4}      # 9.4e-4 passed round 5's constant 9.5e-4; d^2 + e_full^2 says no
    run, calls = _runner(errs, ids)
    out0 = run(cur); calls.clear()
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        out, kept = L.check(run, ids, out0, cur)
    assert kept == lv[4] and calls == [PL.SPLIT_ALL, lv[3], lv[4]]      # full split first, then the rungs above the current one
    assert len([x for x in w if "gdf verify" in str(x.message)]) == 1 and "outside the statistics" not in str(w[-1].message)
    assert L.log == [(tuple(ids), {lv[2]: pytest.approx(1.10e-3, rel=1e-3), lv[3]: pytest.approx(9.4e-4, rel=1e-3), lv[4]: pytest.approx(9.2e-4, rel=1e-3)}, lv[4])]
    assert tuple(ids) in L.verified and L.escalated[tuple(ids)] == lv[4]
    # from then on: the escalation OR-ed with what the table picks for THIS call — never less than either
    assert L.split_for(ids) == lv[4] | PL.choose_split(cfg, ids) == lv[4]
    assert L.split_for(ids, lat=16) & lv[4] == lv[4]
    assert L.split_for(["up-level1-repeat2-res-out"]) == PL.choose_split(cfg, ["up-level1-repeat2-res-out"])      # another layer set: untouched


def test_benign_weights_no_warning_and_vae_out_is_checked_on_the_model_output():
    L = PL.VerifyLadder(ARCH_CONFIGS["1-5"])
    ids = ["up-level2-repeat2-res-out", "vae-out"]                      # 'vae-out' is not a hook of the denoiser (round-6 KeyError)
    run, calls = _runner({0: 3e-4}, ["up-level2-repeat2-res-out"])
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        out, kept = L.check(run, ids, run(0), 0)
    assert kept == 0 and not w and L.log[0][1] == {0: pytest.approx(3e-4, rel=1e-3)} and not L.escalated
    # the model output decides when it is the worse one
    noise_only = PL.VerifyLadder(ARCH_CONFIGS["1-5"])

    def run2(mask):
        n = torch.ones(2, 4, 4, 4)
        return (n if mask == PL.SPLIT_ALL else n * (1 + 2e-3)), {"up-level2-repeat2-res-out": torch.ones(2, 8, 4, 4)}
    with warnings.catch_warnings(record=True):
        warnings.simplefilter("always")
        _, kept = noise_only.check(run2, ids, run2(0), 0)
    assert kept == PL.SPLIT_ALL and noise_only.log[0][1][0] == pytest.approx(2e-3, rel=1e-3)


def test_every_rung_fails_reaches_the_full_split_and_says_so():
    L = PL.VerifyLadder(ARCH_CONFIGS["xl"])
    ids = ["mid-vit-block3-ffn-inner"]
    lv = L.levels()
    run, calls = _runner({m: 2e-3 for m in lv[:-1]}, ids)
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        out, kept = L.check(run, ids, run(lv[2]), lv[2])
    assert kept == PL.SPLIT_ALL and "outside the statistics" in str(w[-1].message)
    assert torch.equal(out[1][ids[0]], run(PL.SPLIT_ALL)[1][ids[0]])    # the full split's own result is handed out


def test_a_check_that_cannot_run_leaves_the_layer_set_unverified():
    L = PL.VerifyLadder(ARCH_CONFIGS["xl"])
    ids = ["mid-vit-block3-ffn-inner"]
    run, calls = _runner({}, ids, fail=(PL.SPLIT_ALL,))
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        out, kept = L.check(run, ids, "unchanged", 4491)
    assert out == "unchanged" and kept == 4491 and tuple(ids) not in L.verified and not L.log
    assert "verify: skipped" in str(w[-1].message)
    # a rung whose plan does not exist is skipped, not fatal
    L2 = PL.VerifyLadder(ARCH_CONFIGS["xl"]); lv = L2.levels()
    run, calls = _runner({lv[2]: 2e-3, lv[4]: 5e-4}, ids, fail=(lv[3],))
    with warnings.catch_warnings(record=True):
        warnings.simplefilter("always")
        _, kept = L2.check(run, ids, run(lv[2]), lv[2])
    assert kept == lv[4] and lv[3] not in L2.log[0][1]


def test_bound_override_and_masks(monkeypatch):
    L = PL.VerifyLadder(ARCH_CONFIGS["xl"])
    L.bound_override = 3e-3
    assert L.accept_bound() == 3e-3
    monkeypatch.setenv("GDF_VERIFY_BOUND", "5e-4")
    assert L.accept_bound() == 5e-4
    assert PL.split_mask("selective") == PL.SPLIT_SELECTIVE and PL.split_mask(True) == PL.SPLIT_ALL and PL.split_mask("plain") == 0
    assert PL.split_mask("stream,attn_out") == PL.SPLIT_CLASSES["stream"] | PL.SPLIT_CLASSES["attn_out"]
    with pytest.raises(ValueError):
        PL.split_mask("no-such-class")
    # the headline's four hooks select the plain plan; anything with ffn-inner / unet-out / a map the selective preset; 'vae-out' counts as unet-out
    four = ["up-level0-repeat0-vit-block7-out", "up-level0-repeat0-vit-block5-out", "up-level1-repeat0-vit-block0-cross-q", "up-level1-repeat0-vit-block0-out"]
    assert PL.choose_split(ARCH_CONFIGS["xl"], four) == 0
    assert PL.choose_split(ARCH_CONFIGS["xl"], four + ["vae-out"]) == PL.choose_split(ARCH_CONFIGS["xl"], four + ["unet-out"]) != 0
    assert PL.choose_split(ARCH_CONFIGS["xl"], ["down-level1-repeat0-vit-block0-self-map"]) == PL.SELECTIVE_BY_ARCH["xl"]
