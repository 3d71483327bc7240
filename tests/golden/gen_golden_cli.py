"""Golden files of the OUTPUT STAGE from the REFERENCE's own code (build container only).

    python tests/golden/gen_golden_cli.py        # needs /root/reference; writes tests/golden/cli_output_stage.npz

The reference's output stage is inline in `main()` of /root/reference/extract_feature.py (:112-148: `--aggregate_output`
nearest-resize + channel concat + one .npy per sample; per-layer and --sample_name_first layouts).  This script cuts exactly
those source lines out of the reference file AT RUN TIME, executes them on seeded synthetic feature tensors and records the
files they wrote.  The fixture is pure data: the input feature tensors (fp16) and, per mode, {relative path: array}."""
import os
import sys
import tempfile
import textwrap
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference/extract_feature.py"


def reference_block():
    src = open(REF).read().split("\n")
    a = next(i for i, l in enumerate(src) if l.strip() == "# save the results")
    b = next(i for i, l in enumerate(src) if i > a and l.strip() == "np.save(out_path, feat)")
    return textwrap.dedent("\n".join(src[a:b + 1]))


def features():
    g = torch.Generator().manual_seed(0)
    return {"up-level1-repeat2-res-out": torch.randn(3, 6, 4, 4, generator=g).half(),
            "up-level3-repeat0-vit-block0-self-k": torch.randn(3, 5, 8, 8, generator=g).half(),
            "mid-vit-block0-cross-q": torch.randn(3, 3, 2, 2, generator=g).half(),
            "up-level2-repeat1-vit-block0-cross-q": torch.randn(3, 4, 6, 6, generator=g).half()}      # 6 -> 8: non-integer nearest ratio


MODES = {
    "aggregate": dict(aggregate_output=True, use_original_filename=True, nested_input_dir=False, sample_name_first=False, split="train"),
    "aggregate_nested": dict(aggregate_output=True, use_original_filename=True, nested_input_dir=True, sample_name_first=False, split="train"),
    "per_layer": dict(aggregate_output=False, use_original_filename=False, nested_input_dir=False, sample_name_first=False, split="val"),
    "sample_first": dict(aggregate_output=False, use_original_filename=True, nested_input_dir=False, sample_name_first=True, split="train"),
}
NAMES = {False: ["a", "b", "c"], True: ["d0/a", "d0/b", "d1/c"]}


def main():
    code = compile(reference_block(), REF, "exec")
    feats = features()
    arrs = {"feat:" + k: v.numpy() for k, v in feats.items()}
    for mode, kw in MODES.items():
        with tempfile.TemporaryDirectory() as d:
            args = types.SimpleNamespace(output_dir=d, **kw)
            names = NAMES[kw["nested_input_dir"]]
            ns = dict(args=args, features=feats, sublist=[None] * 3, target_dataset=[(None, None, n) for n in names], i=0,
                      np=np, torch=torch, os=os)
            exec(code, ns)
            for root, _, files in os.walk(d):
                for f in files:
                    rel = os.path.relpath(os.path.join(root, f), d)
                    arrs[f"file:{mode}:{rel}"] = np.load(os.path.join(root, f))
    arrs["meta"] = np.array(repr(dict(modes=MODES, names=NAMES, order=list(feats.keys()))))
    path = os.path.join(HERE, "cli_output_stage.npz")
    np.savez_compressed(path, **arrs)
    print(len([k for k in arrs if k.startswith("file:")]), "files ->", os.path.getsize(path) // 1024, "KiB")


if __name__ == "__main__":
    main()
