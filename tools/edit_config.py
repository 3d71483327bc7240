#!/usr/bin/env python3
"""Bulk-edit a layer-selection config ({layer_id: bool}): switch every id that contains one of the given substrings on or off.
Same job as the reference's feature/configs/edit_config.py (which hard-codes its file names and substrings), as a command line:

    python tools/edit_config.py generic-diffusion-feature_amd/configs/config_xl_full.json out.json --off down mid
    python tools/edit_config.py in.json out.json --only up-level1 --off self-map cross-map   # keep up-level1 ids, without attention maps
"""
import argparse
import json

ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
ap.add_argument("src"); ap.add_argument("dst")
ap.add_argument("--on", nargs="*", default=[], help="substrings: matching ids become true")
ap.add_argument("--off", nargs="*", default=[], help="substrings: matching ids become false (applied after --on)")
ap.add_argument("--only", nargs="*", default=None, help="substrings: ids matching NONE of them become false first")
a = ap.parse_args()
with open(a.src) as f:
    cfg = json.load(f)                                     # key order = hook execution order: preserved
n0 = sum(bool(v) for v in cfg.values())
for k in cfg:
    if a.only is not None and not any(s in k for s in a.only):
        cfg[k] = False
    if any(s in k for s in a.on):
        cfg[k] = True
    if any(s in k for s in a.off):
        cfg[k] = False
with open(a.dst, "w") as f:
    json.dump(cfg, f, indent=1)
    f.write("\n")
print(f"{a.src}: {n0} of {len(cfg)} ids selected -> {a.dst}: {sum(bool(v) for v in cfg.values())}")
