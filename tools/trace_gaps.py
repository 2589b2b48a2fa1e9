#!/usr/bin/env python3
"""Idle time BETWEEN consecutive kernels of a rocprofv3 kernel trace (the unfused step's three launches).
    python3 tools/trace_gaps.py <dir with *_kernel_trace.csv> [name-substring ...]
For every pair of consecutive dispatches (by start time) whose names both match one of the substrings (default: the three
kernels of bench.py's unfused step) prints the distribution of  start(next) - end(previous)  and, per kernel, its duration inside
the alternating sequence next to its duration when it follows itself (the stand-alone timing loops of bench.py)."""
import csv
import glob
import os
import sys
from collections import defaultdict

root = sys.argv[1]
keys = sys.argv[2:] or ["ntt_kernel_wp<unsigned long, 10", "pointwise_kernel<unsigned long, 0>"]
rows = []
for f in glob.glob(os.path.join(root, "**", "*kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
short = lambda n: n.replace("cntt::", "").replace("unsigned long", "u64").split("(")[0][:60]
gaps, dur_alt, dur_self = defaultdict(list), defaultdict(list), defaultdict(list)
for (s0, e0, n0), (s1, e1, n1) in zip(rows, rows[1:]):
    if not (any(k in n0 for k in keys) and any(k in n1 for k in keys)):
        continue
    if n0 != n1:
        gaps[(short(n0), short(n1))].append((s1 - e0) / 1e3)
        dur_alt[short(n1)].append((e1 - s1) / 1e3)
    else:
        dur_self[short(n1)].append((e1 - s1) / 1e3)


def stats(v):
    v = sorted(v)
    return "n=%5d  median %8.2f us  p90 %8.2f  max %8.2f" % (len(v), v[len(v) // 2], v[int(len(v) * 0.9)], v[-1])


print("# gaps between DIFFERENT consecutive kernels (start of the next - end of the previous)")
for k, v in sorted(gaps.items()):
    print("%-50s -> %-50s %s" % (k[0], k[1], stats(v)))
print("# kernel duration when it follows a different kernel / when it follows itself")
for k in sorted(set(dur_alt) | set(dur_self)):
    print("%-60s alternating: %s" % (k, stats(dur_alt[k]) if dur_alt[k] else "-"))
    print("%-60s back to back: %s" % ("", stats(dur_self[k]) if dur_self[k] else "-"))
