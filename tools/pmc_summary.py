#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc csv output (counter_collection.csv): mean per-dispatch counter values per kernel."""
import csv, glob, sys, collections, re
files = glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True)
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in files:
    for r in csv.DictReader(open(f)):
        k = re.sub(r"\(.*", "", r["Kernel_Name"])[:60] + f" grid={r.get('Grid_Size','')}"
        agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in agg.items():
    if "gdf" not in k: continue
    print(k)
    for c, v in sorted(d.items()):
        print(f"    {c:32s} n={len(v):3d} mean={sum(v)/len(v):16.1f}")
