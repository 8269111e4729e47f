#!/usr/bin/env python3
"""Per-dispatch statistics of a rocprofv3 --kernel-trace database (rocpd .db): for every kernel the mean / median / min / p10 / p90
of its dispatch durations, and for the dominant kernel the PERIOD between consecutive dispatches (start to start — what a wall clock
or a pair of HIP events around the region divides by the number of launches) and how often a dispatch starts before its predecessor
ended (the profiler's begin / end stamps of back-to-back dispatches overlap: the next dispatch's ramp over the XCDs runs under the
previous one's tail).  Usage: kt_stats.py <results.db> [kernel-substring]"""
import sqlite3
import statistics
import sys

db, flt = sys.argv[1], (sys.argv[2] if len(sys.argv) > 2 else "step_kernel")
con = sqlite3.connect(db)
rows = con.execute("select name, start, end from kernels order by start").fetchall()
by = {}
for n, s, e in rows:
    by.setdefault(n.split("(")[0], []).append((s, e))
print(f"# {db}")
print("| kernel | calls | mean us | median | min | p10 | p90 | max |")
print("|---|---|---|---|---|---|---|---|")
q = lambda v, p: sorted(v)[min(len(v) - 1, int(p * len(v)))]
for n, se in sorted(by.items(), key=lambda kv: -sum(e - s for s, e in kv[1])):
    d = [(e - s) / 1e3 for s, e in se]
    print(f"| {n} | {len(d)} | {statistics.fmean(d):.3f} | {statistics.median(d):.3f} | {min(d):.3f} | {q(d, .1):.3f} | {q(d, .9):.3f} | {max(d):.3f} |")
step = [(n, se) for n, se in by.items() if flt in n]
if step:
    allse = sorted(x for _, se in step for x in se)
    per, gap, ov = [], [], 0
    for (s0, e0), (s1, e1) in zip(allse, allse[1:]):
        if s1 - e0 < 20000:  # same region (a gap of > 20 us is a host-side break between regions)
            per.append((s1 - s0) / 1e3)
            gap.append((s1 - e0) / 1e3)
            ov += s1 < e0
    if per:
        print(f"\n`{flt}` dispatches back to back ({len(per)} pairs): period start-to-start mean {statistics.fmean(per):.3f} us, median "
              f"{statistics.median(per):.3f}, min {min(per):.3f}; gap end-to-next-start mean {statistics.fmean(gap):.3f} us, median "
              f"{statistics.median(gap):.3f}; {100.0 * ov / len(per):.1f} % of the dispatches start before their predecessor's end stamp")
    # a sample of consecutive dispatches from the middle of the run: duration / gap to the next start (us) — is the spread a pattern?
    mid = len(allse) // 2
    print("\nconsecutive dispatches (duration | gap to the next start), 48 from the middle of the trace:")
    print("  " + "  ".join(f"{(e0 - s0) / 1e3:.2f}|{(s1 - e0) / 1e3:.2f}" for (s0, e0), (s1, e1) in zip(allse[mid:mid + 48], allse[mid + 1:mid + 49])))
    d = sorted((e - s) / 1e3 for s, e in allse)
    edges = [5.0, 5.5, 6.0, 6.5, 7.0, 7.5, 8.0, 8.5, 9.0, 9.5, 10.0, 11.0, 12.0, 1e9]
    hist, lo = [], 0.0
    for hi in edges:
        hist.append((lo, hi, sum(1 for x in d if lo <= x < hi)))
        lo = hi
    print("duration histogram (us: count): " + "  ".join(f"[{lo:g},{'inf' if hi > 1e8 else format(hi, 'g')}): {n}" for lo, hi, n in hist))
