#!/usr/bin/env python3
"""Aggregate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes (csv) into per-launch HBM traffic per kernel.
    python tools/pmc_traffic.py <fetch_dir> <write_dir> > profiles/r01_pmc_traffic.json
Corrections (MI355X_MICROARCH.md, HBM): counters are in KiB; on gfx950 FETCH_SIZE reports half of the bytes of wide
(16 B/lane) coalesced reads -> x2 for the read side (all hot kernels here read with 16-B lanes / LDS-DMA)."""
import csv, glob, json, re, sys, collections


def load(d, counter):
    agg = collections.defaultdict(list)
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == counter:
                agg[re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void gdf::", "")].append(float(r["Counter_Value"]))
    return agg


fe, wr = load(sys.argv[1], "FETCH_SIZE"), load(sys.argv[2], "WRITE_SIZE")
out = {}
for k in fe:
    if "gemm_kernel" not in k and "attn_kernel" not in k and "gdf" not in k:
        continue
    f = sum(fe[k]) / len(fe[k]) * 1024 * 2.0
    w = sum(wr.get(k, [0])) / max(1, len(wr.get(k, [0]))) * 1024
    out[k] = dict(launches=len(fe[k]), fetch_bytes_per_launch=round(f), write_bytes_per_launch=round(w),
                  hbm_bytes_per_launch=round(f + w))
print(json.dumps(out, indent=1))
