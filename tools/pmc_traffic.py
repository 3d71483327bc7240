#!/usr/bin/env python3
"""Aggregate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes (csv) into per-launch HBM traffic per kernel.
    python tools/pmc_traffic.py <fetch_dir> <write_dir> > profiles/pmc_traffic_current.json
Corrections (MI355X_MICROARCH.md, HBM): counters are in KiB; on gfx950 FETCH_SIZE reports half of the bytes of wide
(16 B/lane) coalesced reads -> x2 for the read side (all hot kernels here read with 16-B lanes / LDS-DMA)."""
import csv, glob, json, os, re, sys, collections, time


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
# stamp: bench.py only reports `roofline.traffic` from a file whose csrc_sha equals the hash of the sources it runs on
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
out["_meta"] = dict(csrc_sha=bench.csrc_sha(), taken=time.strftime("%Y-%m-%d %H:%M:%S"),
                    command="rocprofv3 --pmc FETCH_SIZE | WRITE_SIZE (separate passes) -- python3 bench.py --steps 2 --warmup 1",
                    corrections="KiB -> bytes; FETCH_SIZE x2 (gfx950 counts 128-B requests as 64 B: MI355X_MICROARCH.md, HBM)")
print(json.dumps(out, indent=1))
