#!/usr/bin/env python3
"""Summarise a rocprofv3 (ROCm 7.2, rocpd sqlite) kernel trace into a --stats style table.

    python tools/rocpd_stats.py gpurun_out/prof_r1/bench_results.db > profiles/r01_bench_kernel_stats.txt
"""
import re
import sqlite3
import sys


def short(n):
    n = re.sub(r"\(.*$", "", n)
    return n if len(n) < 110 else n[:107] + "..."


def main(path):
    c = sqlite3.connect(path)
    cur = c.cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
    name_col = "name" if "name" in cols else [x for x in cols if "name" in x][0]
    rows = cur.execute(f"select {name_col}, start, end from kernels").fetchall()
    agg = {}
    for n, s, e in rows:
        d = agg.setdefault(n, [0, 0, 10 ** 18, 0])
        dt = e - s
        d[0] += 1; d[1] += dt; d[2] = min(d[2], dt); d[3] = max(d[3], dt)
    tot = sum(v[1] for v in agg.values())
    print(f"# rocprofv3 --kernel-trace summary of {path}: {len(rows)} dispatches, {tot / 1e6:.3f} ms total kernel time")
    print(f"{'Name':110s} {'Calls':>7s} {'TotalDuration(ns)':>18s} {'Average(ns)':>12s} {'Percentage':>10s} {'Min(ns)':>10s} {'Max(ns)':>10s}")
    for n, v in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        print(f"{short(n):110s} {v[0]:7d} {v[1]:18d} {v[1] / v[0]:12.0f} {100.0 * v[1] / tot:10.2f} {v[2]:10d} {v[3]:10d}")


if __name__ == "__main__":
    main(sys.argv[1])
