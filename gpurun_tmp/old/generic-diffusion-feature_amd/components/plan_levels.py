"""Operand-plan levels of the UNet path: which activation operands of a plan are kept as split fp16 pairs (hi + lo), how the cheapest
level that keeps every REQUESTED hook within 1e-3 of the fp32 reference is chosen from the emulated per-hook error table
(`choose_split`), and the runtime self-check that measures the chosen level against the full split on the caller's own weights and
climbs a ladder of levels when it is not enough (`VerifyLadder`).

Split out of components/native.py in round 6: this module is pure policy — no ctypes, no device memory; native.py
binds it to plans.  Reference contract being protected: FeatureStore hands out what the model computed
(/root/reference/feature/components/feature_extractor.py:31-76); here: within 1e-3 relative L2 of it.
"""
import os

_HERE = os.path.dirname(os.path.abspath(__file__))

# Operand classes of a UNet plan that can be kept as split fp16 pairs hi + lo (csrc/builder.h SP_*, include/gdf.h gdf_plan_opts.reserved[1]).
SPLIT_CLASSES = {"stream": 1, "gnv": 2, "ln_attn": 4, "attn_out": 8, "ln_ff": 16, "ff_inner": 32, "res": 64, "out": 128, "sampler": 256,
                 "attn2_out": 512, "upsampler": 1024,
                 # round 5: the self-attention q | k | v stored as pairs, the flash kernel contracting over both halves (csrc/attn.hip QKP): the storage
                 # rounding in front of the softmax — the floor of the full split on heavy-tailed weight statistics (DESIGN.md 3.9 h)
                 "qkv": 2048,
                 # ... and of the text cross-attention (its q, the grouped text K / V): 77 keys, < 1 % of the step, and by far the larger half of that
                 # rounding (benign 4.0e-4 of the worst hook against 0.7e-4 for the self-attention; heavy-tailed 8.0e-4 against 2.6e-4)
                 "xqkv": 4096}
SPLIT_ALL = 8191
# The SELECTIVE preset: the main-path roundings (fp16 images of the residual stream as read by the shortcuts, proj_out, the GroupNorms and the
# DOWNsampler convs; the GroupNorm output in front of proj_in; conv_out's operand) plus the SELF-attention outputs — the cheapest subset whose
# CPU-emulated worst hook stays below 8.5e-4 on SDXL and SD1.5 (tools/operand_subsets.py, profiles/r04_operand_subsets_*).  Not in it although
# on the list of candidates: the cross-attention outputs (1e-8 of variance) and the two upsampler convs (4.5e-8 for 3 ms of the step).
# Round 5: + the cross-attention q / k / v pairs (`xqkv`): 4.0e-4 of the worst hook on benign weights for < 1 % of the step (the committed table
# was emulated without it: its selective / light columns are upper bounds).
SPLIT_SELECTIVE = (SPLIT_CLASSES["stream"] | SPLIT_CLASSES["gnv"] | SPLIT_CLASSES["attn_out"] | SPLIT_CLASSES["out"] | SPLIT_CLASSES["sampler"] | SPLIT_CLASSES["xqkv"])
# per architecture family: SD1.5 / SD2.1 (one transformer block per level) do not need the attention outputs
# The LIGHT level (round 5): only the `gnv` class — the GroupNorm output in front of proj_in as an fp16 pair (one small GEMM per transformer with K
# doubled; < 1 % of the step).  It removes the proj_in operand rounding, 3.1-3.3e-4 of every later hook's error on SD1.5: enough for the hooks whose
# plain-plan error sits just above the bound (SD1.5's practical `self-k`: 9.7e-4 plain, 9.1e-4 light) at a fraction of the selective preset's cost.
SPLIT_LIGHT = SPLIT_CLASSES["gnv"] | SPLIT_CLASSES["xqkv"]
SELECTIVE_BY_ARCH = {"xl": SPLIT_SELECTIVE, "1-5": SPLIT_CLASSES["stream"] | SPLIT_CLASSES["gnv"] | SPLIT_CLASSES["out"] | SPLIT_CLASSES["sampler"] | SPLIT_CLASSES["xqkv"]}
# The DEEP levels (round 5, only on verify's ladder, between the selective preset and the full split): selective + the GEGLU projection's operand,
# then + the self-attention q / k / v pairs as well.  On heavy-tailed weight statistics the error is spread over every class, and the cheapest subset the per-class emulation
# finds under 8e-4 is this one (tools/operand_subsets.py --heavy with the qkv class: 7.3e-4 at +34 % of the step, where the full split costs +75 %;
# profiles/r05_heavy_tailed_operand_classes.txt).  The table never selects it: a layer set reaches it only when verify measured that the selective
# preset is not enough for THESE weights.
SPLIT_DEEP_EXTRA = SPLIT_CLASSES["qkv"] | SPLIT_CLASSES["ln_ff"]
# emulated error above which the next plan level is chosen.  Round 5, measured at full size on hardware (tests/test_gpu_fullsize.py
# _plain_plan_contract, profiles/r05_plain_plan_contract.txt): hooks whose emulated error is within 15 % of the bound measure 1.3-4.3 % (SDXL B = 16) /
# 0-4.5 % (SD1.5) above the emulation; at 9.25e-4 the worst hook handed to the plain plan measures 9.5e-4 (SDXL) / 9.6e-4 (SD1.5): >= 4 % below 1e-3
# for EVERY hook a caller may request alone (asserted per hook), and bound x worst offset (1.045) = 9.67e-4.  (9.5e-4 accepted
# mid-vit-block2-ffn-inner at 9.70e-4: 3.0 %.)
AUTO_BOUND = 9.25e-4
# ... and per family, because the error of a hook also moves with the INPUTS (the table was emulated on one seeded sample / prompt at t = 100): measured at true
# widths on the plain plan for three samples / prompts x t in {20, 100, 500, 900} plus a two-sample batch (tools/input_variation.py, profiles/r05_input_variation.txt,
# tests/test_gpu_fullsize.py::test_sdxl_plan_level_contract_on_other_inputs) a hook's worst case sits up to 12-14 % above the table on the SDXL family (ten
# transformer blocks per level: p90 1.09) but only up to 7 % on the SD1.5 family (one block per level).  Bounds under which the worst hook handed to the plain plan
# over ALL those inputs measures 9.45e-4 (SDXL) / 9.48e-4 (SD1.5): >= 5 % of headroom for inputs nobody has tried.
AUTO_BOUND_BY_FAMILY = {"xl": 8.3e-4, "1-5": 9.1e-4}


def auto_bound(cfg):
    return AUTO_BOUND_BY_FAMILY.get(arch_family(cfg), AUTO_BOUND)
_ERR_TABLE = None


def arch_family(cfg):
    """'xl' (3 levels, deep transformers: SDXL / Playground-v2), '1-5' (4 levels, one block per level: SD1.5 / SD2.1) or None (anything else)"""
    boc, tl = tuple(cfg["block_out_channels"]), tuple(cfg["transformer_layers"])
    if boc == (320, 640, 1280) and tl[:3] == (1, 2, 10):
        return "xl"
    if boc == (320, 640, 1280, 1280) and set(tl) == {1}:
        return "1-5"
    return None


RES_EXPONENT = 0.13      # error ~ (table resolution / resolution)^0.13 below the table's resolution (see choose_split)


def table_scale(cfg, lat=None):
    """Factor applied to the operand-error table for THIS model and latent grid (>= 1; choose_split compares table x factor with AUTO_BOUND)."""
    global _ERR_TABLE
    fam = arch_family(cfg)
    if _ERR_TABLE is None:
        import json
        try:
            _ERR_TABLE = json.load(open(os.path.join(_HERE, "operand_error_table.json")))
        except Exception:
            _ERR_TABLE = {}
    scale = 1.0
    if fam == "1-5" and (tuple(cfg.get("heads", ())) != (8, 8, 8, 8) or cfg.get("cross_attention_dim") != 768):
        # SD2.1-base borrows the SD1.5 table (same topology, 64-wide heads, linear projections, cross dim 1024): measured at true widths its hooks
        # sit up to 7.8 % above the table where SD1.5's own sit 4.5 % above (tests/test_gpu_fullsize.py::test_sd21_512_plan_levels_borrow_the_sd15_table)
        scale = 1.04
    if lat and fam:
        lat_t = float(_ERR_TABLE.get(fam, {}).get("lat", 0) or 0)
        if lat_t > 0 and lat < lat_t:
            scale *= (lat_t / float(lat)) ** RES_EXPONENT
    return scale


def choose_split(cfg, hook_ids, lat=None):
    """The cheapest operand-class mask under which every requested hook stays within BASELINE.json's 1e-3 of the fp32 reference:
    0 (plain fp16 operands) -> SPLIT_LIGHT -> the architecture's selective preset -> SPLIT_ALL.  Decided from components/operand_error_table.json (per-hook
    error of the CPU oracle with exactly the plan's operand classes rounded to fp16, tools/operand_subsets.py): a level is accepted when
    every requested hook's emulated error is <= AUTO_BOUND.  `*-map` hooks (not in the table: too large to emulate per class) and unknown
    architectures / hook ids fall back to kind rules: maps, `ffn-inner`, `unet-out` need the selective preset."""
    global _ERR_TABLE
    ids = [h for h in hook_ids]
    if not ids:
        return 0
    fam = arch_family(cfg)
    if _ERR_TABLE is None:
        import json
        try:
            _ERR_TABLE = json.load(open(os.path.join(_HERE, "operand_error_table.json")))
        except Exception:
            _ERR_TABLE = {}
    tab = _ERR_TABLE.get(fam, {}).get("hooks") if fam else None
    sel = SELECTIVE_BY_ARCH.get(fam, SPLIT_SELECTIVE)
    # The table was emulated at the BASELINE resolution (latent 128 for the SDXL family, 64 for SD1.5).  Smaller grids average the operand
    # rounding over fewer elements: measured on hardware (tests/test_gpu_fullsize.py ragged-shape test) the selective plan's worst hook goes
    # 8.2e-4 -> 9.1e-4 from 1024^2 to 448^2 (x 1.11 for 2.29x fewer rows per side), the plain plan's median 7.8e-4 -> 8.2e-4: the table values are
    # scaled by (table lat / lat)^0.13 when the call's latent grid is smaller (never down-scaled for larger grids).
    scale = table_scale(cfg, lat)
    bound = auto_bound(cfg) / scale
    level = 0                                            # 0 plain, 1 light, 2 selective, 3 full
    for h in ids:
        if h.endswith("-map"):
            level = max(level, 2)
            continue
        if h == "vae-out":                               # decode(step(latents, noise_pred)): as accurate as the noise prediction
            h = "unet-out"
        row = tab.get(h) if tab else None
        if row is None:                                  # unknown architecture or id: conservative kind rule
            risky = h.endswith(("ffn-inner", "unet-out", "-q", "-k", "-v")) or (fam != "xl" and h.endswith(("-out", "res-increment")))
            level = max(level, 2 if risky else 0)
            continue
        if row[0] <= bound:
            continue
        if len(row) > 2 and row[2] <= bound:             # columns: plain, selective, light
            level = max(level, 1)
            continue
        level = max(level, 2 if row[1] <= bound else 3)
        if level == 3:
            break
    return (0, SPLIT_LIGHT, sel, SPLIT_ALL)[level]


def split_mask(spec):
    """None / False / 0 -> 0 (plain fp16 operands); True / 'precise' / 'all' -> every class; 'selective' -> SPLIT_SELECTIVE;
    'stream,attn_out' -> those classes; an int is taken as the mask itself."""
    if spec is None or spec is False:
        return 0
    if spec is True:
        return SPLIT_ALL
    if isinstance(spec, int):
        return spec & SPLIT_ALL
    m = 0
    for tok in str(spec).replace("+", ",").split(","):
        tok = tok.strip().lower()
        if not tok or tok in ("0", "none", "default", "plain"):
            continue
        if tok in ("1", "all", "precise", "full"):
            m |= SPLIT_ALL
        elif tok == "selective":
            m |= SPLIT_SELECTIVE
        elif tok == "light":
            m |= SPLIT_LIGHT
        elif tok == "deep":
            m |= SPLIT_SELECTIVE | SPLIT_DEEP_EXTRA
        elif tok in SPLIT_CLASSES:
            m |= SPLIT_CLASSES[tok]
        else:
            raise ValueError(f"unknown split-operand class {tok!r}; known: {sorted(SPLIT_CLASSES)} + 'light', 'selective', 'deep', 'precise'")
    return m


class VerifyLadder:
    """State + policy of the runtime self-check behind `verify=True` / GDF_VERIFY=1 of a NativeUNet (one per model).

    `check(run, ids, out, cur, drop)`: `run(mask)` -> (noise, hooks) of the SAME inputs under operand mask `mask`; `out` = the result of the level
    `cur` the table chose; returns (result to hand out, level kept).  Compares every requested hook with the FULL split (itself 1.9-4.4e-4 from
    fp32 on benign AND heavy-tailed weights, tests/test_gpu_fullsize.py) and climbs
        plain -> light -> selective -> selective + GEGLU operand -> deep (+ q / k / v pairs) -> every transformer-side class
              -> everything but the ResBlock conv operands -> full
    until the worst relative L2 difference d satisfies d^2 + e_full^2 <= TARGET^2 (e_full = the full split's own distance to fp32 for the
    family; the two are independent roundings).  What the check cannot see is e_full itself: until round 5 that was 0.7-1.5e-3 on the synthetic
    heavy-tailed statistics of oracle/unet_ref.py synth_params_heavy (the fp16 STORAGE of q / k / v in front of the text cross-attention's
    peaked softmax); the full split now carries q / k / v of both attentions as pairs too (csrc/attn.hip QKP) and measures 1.9-2.7e-4 on the
    true SDXL widths (DESIGN.md 3.9 h).  Reaching the full split is reported as "outside the statistics the plan table was built on"."""

    TARGET = 0.97e-3                                              # what a kept level must meet against fp32: 3 % under the north star's 1e-3
    FULL_SPLIT_ERROR = {"xl": 2.7e-4, "1-5": 4.4e-4}              # the full split's own measured distance to fp32, worst hook incl. maps

    def __init__(self, cfg):
        self.cfg = cfg
        self.bound_override = None       # a float replaces the arithmetic bound (tests)
        self.escalated = {}              # tuple(hook ids) -> split mask found necessary
        self.verified = set()            # layer sets whose comparison COMPLETED
        self.log = []                    # [(hook ids tuple, {mask: worst difference to the full split}, mask kept)]

    def accept_bound(self):
        env = os.environ.get("GDF_VERIFY_BOUND", "")
        if env:
            return float(env)
        if self.bound_override is not None:
            return float(self.bound_override)
        e_full = self.FULL_SPLIT_ERROR.get(arch_family(self.cfg), 4.4e-4)
        return (self.TARGET ** 2 - e_full ** 2) ** 0.5

    def levels(self):
        sel = SELECTIVE_BY_ARCH.get(arch_family(self.cfg), SPLIT_SELECTIVE)
        deep = sel | SPLIT_DEEP_EXTRA
        return [0, SPLIT_LIGHT, sel, sel | SPLIT_CLASSES["ln_ff"], deep,
                # round 6: two more rungs below the full split — every transformer-side class, then everything but the
                # ResBlock conv operands (`res`: the most expensive class, a third of an SD1.5 step) — so that weights whose error is spread over
                # all classes (the heavy-tailed SD1.5 draw: 1.1-1.5e-3 on every rung up to `deep`, 8.2e-4 on the first new one) do not fall
                # straight to the full split
                deep | SPLIT_CLASSES["attn_out"] | SPLIT_CLASSES["ff_inner"] | SPLIT_CLASSES["ln_attn"] | SPLIT_CLASSES["attn2_out"],
                SPLIT_ALL & ~SPLIT_CLASSES["res"], SPLIT_ALL]

    def split_for(self, hook_ids, lat=None):
        """the table's choice for this grid OR-ed with what verify found necessary for this layer set (the masks are nested, and a
        smaller grid / another batch may need more than the verified one did)"""
        return self.escalated.get(tuple(hook_ids), 0) | choose_split(self.cfg, hook_ids, lat)

    def check(self, run, ids, out, cur):
        import warnings
        key = tuple(ids)
        if cur == SPLIT_ALL or not ids:
            self.verified.add(key)
            return out, cur
        levels = self.levels()
        ladder = levels[levels.index(cur):] if cur in levels else [cur] + [m for m in levels if (m & cur) == cur and m != cur]
        try:
            ref = run(SPLIT_ALL)
        except RuntimeError as e:
            # the reference plan does not exist at this size (32-bit buffer offsets: the full split halves the largest batch) or does not fit in
            # memory: the check cannot run; say so and keep the table's choice.  The layer set stays UNVERIFIED: the next forward of
            # it — e.g. on the smaller batch the warning asks for — tries again.
            warnings.warn(f"gdf verify: skipped for this layer set ({str(e)[:120]}); verify on a smaller batch to check the automatic operand plan "
                          "against these weights", RuntimeWarning, stacklevel=4)
            return out, cur
        bound = self.accept_bound()
        seen = {}
        for m in ladder:
            if m == SPLIT_ALL:
                out = ref
                break
            if m != cur:
                try:
                    out = run(m)
                except RuntimeError:
                    continue
            worst = 0.0
            for k in ids:
                if k == "vae-out":                    # not a hook of the denoiser: decode(step(latents, noise_pred)) — as accurate as the model output
                    a, r = out[0].float(), ref[0].float()
                elif k in ref[1]:
                    a, r = out[1][k].float(), ref[1][k].float()
                else:
                    continue
                worst = max(worst, float((a - r).norm() / (r.norm() + 1e-30)))
            seen[m] = worst
            if worst <= bound:
                break
        else:
            m = SPLIT_ALL
        kept = m
        self.verified.add(key)                        # only after a COMPLETED comparison
        self.log.append((key, seen, kept))
        if kept != cur:
            self.escalated[key] = kept
            tail = ""
            if kept == SPLIT_ALL:
                tail = ("; every cheaper level differs from the full split by more than the bound: these weights are outside the statistics the plan "
                        "table was built on (the full split itself measures 2-4e-4 against fp32 on benign and on heavy-tailed synthetic weights)")
            warnings.warn(f"gdf verify: operand plan {cur} differs from the full split by {seen.get(cur, float('nan')):.2e} (> {bound:.2e}) on the "
                          f"requested layers with THESE weights; using plan {kept} for this layer set from now on "
                          f"(levels tried: { {k: '%.2e' % v for k, v in seen.items()} }){tail}", RuntimeWarning, stacklevel=4)
        return out, kept
