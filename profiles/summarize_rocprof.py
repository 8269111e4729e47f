#!/usr/bin/env python3
"""Summarise rocprofv3 rocpd (.db) outputs: per-kernel time stats and per-kernel mean PMC values.
Usage: summarize_rocprof.py <results.db> [kernel-substring]"""
import sqlite3
import sys
from collections import defaultdict

db = sys.argv[1]
flt = sys.argv[2] if len(sys.argv) > 2 else ""
con = sqlite3.connect(db)
print(f"# {db}")
try:
    rows = con.execute("select name, total_calls, total_duration, average, percentage from top_kernels").fetchall()
    print("## kernel-trace stats (durations in us)")
    print("| kernel | calls | total_us | avg_us | pct |")
    print("|---|---|---|---|---|")
    for r in rows:
        print(f"| {r[0]} | {r[1]} | {r[2]:.1f} | {r[3]:.3f} | {r[4]:.2f} |")
except sqlite3.Error as e:
    print("no top_kernels:", e)
try:
    acc = defaultdict(lambda: [0.0, 0])
    meta = {}
    for name, ctr, val, vg, sg, lds, grid, wg in con.execute(
            "select kernel_name, counter_name, value, vgpr_count, sgpr_count, lds_block_size, grid_size, workgroup_size "
            "from counters_collection"):
        if flt and flt not in name:
            continue
        a = acc[(name, ctr)]
        a[0] += val
        a[1] += 1
        meta[name] = (vg, sg, lds, grid, wg)
    if acc:
        print("## PMC (mean per dispatch, summed over instances)")
        print("| kernel | counter | mean | dispatches |")
        print("|---|---|---|---|")
        for (name, ctr), (s, n) in sorted(acc.items()):
            print(f"| {name[:60]} | {ctr} | {s / n:.1f} | {n} |")
        for name, m in meta.items():
            print(f"# {name[:60]}: vgpr={m[0]} sgpr={m[1]} lds={m[2]} grid={m[3]} wg={m[4]}")
except sqlite3.Error as e:
    print("no counters:", e)
