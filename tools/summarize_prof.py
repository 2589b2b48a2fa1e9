#!/usr/bin/env python3
"""Condense rocprofv3 CSV output (kernel trace stats + PMC passes) into a short text summary."""
import csv
import glob
import os
import sys
from collections import defaultdict

root = sys.argv[1]


def short(name):
    name = name.replace("cntt::", "").replace("unsigned long", "u64").replace("unsigned int", "u32")
    return name[:110]


print("# kernel trace (durations in us)")
for f in glob.glob(os.path.join(root, "trace", "**", "*kernel_trace.csv"), recursive=True):
    rows = list(csv.DictReader(open(f)))
    agg = defaultdict(list)
    meta = {}
    for r in rows:
        k = r.get("Kernel_Name", "?")
        agg[k].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
        meta[k] = (r.get("VGPR_Count"), r.get("Accum_VGPR_Count"), r.get("SGPR_Count"), r.get("LDS_Block_Size"),
                   r.get("Workgroup_Size"), r.get("Grid_Size"))
    tot = sum(sum(v) for v in agg.values())
    for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
        print("%-110s calls=%4d avg=%9.2f min=%9.2f max=%9.2f total=%10.1f (%5.1f%%) vgpr/agpr/sgpr/lds/wg/grid=%s" % (
            short(k), len(v), sum(v) / len(v), min(v), max(v), sum(v), 100 * sum(v) / tot, meta[k]))
for f in glob.glob(os.path.join(root, "trace", "**", "*kernel_stats.csv"), recursive=True):
    print("# rocprofv3 --stats:", f)
    print(open(f).read()[:3000])

print("\n# PMC (per-dispatch average by kernel)")
for d in sorted(glob.glob(os.path.join(root, "pmc_*"))):
    if not os.path.isdir(d):
        continue
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        rows = list(csv.DictReader(open(f)))
        agg = defaultdict(lambda: defaultdict(list))
        for r in rows:
            agg[r.get("Kernel_Name", "?")][r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, cs in agg.items():
            if "ntt_kernel" not in k and "pointwise" not in k and "mul_kernel" not in k and "split" not in k and "crt" not in k:
                continue
            print(short(k))
            for c, v in cs.items():
                print("    %-24s n=%3d avg=%16.1f" % (c, len(v), sum(v) / len(v)))
