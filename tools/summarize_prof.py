#!/usr/bin/env python3
"""Condense rocprofv3 CSV output (kernel trace stats + PMC passes, tools/profile.sh) into a short text summary, or
(--json) into the HBM-traffic table bench.py reads from profiles/pmc_traffic.json.

HBM bytes per launch = 2 x FETCH_SIZE (gfx950 tallies the 128-B requests of wide streaming reads at 64 B:
MI355X_MICROARCH.md, section HBM) + WRITE_SIZE, both reported in KiB, from separate --pmc passes."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

root = sys.argv[1]
as_json = "--json" in sys.argv[2:]

OURS = ("ntt_kernel", "pointwise", "mul_kernel", "split", "crt", "native_polymul", "ext_kernel", "ext_accumulate",
        "product_", "global_stage", "fill_uniform")


def short(name):
    name = name.replace("cntt::", "").replace("unsigned long", "u64").replace("unsigned int", "u32").replace("void ", "")
    return name.split("(")[0][:96]


def pmc_tables():
    """{kernel: {counter: [values]}} over every pmc_* directory."""
    agg = defaultdict(lambda: defaultdict(list))
    for d in sorted(glob.glob(os.path.join(root, "pmc_*"))):
        if not os.path.isdir(d):
            continue
        for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            for r in csv.DictReader(open(f)):
                k = r.get("Kernel_Name", "?")
                if any(t in k for t in OURS):
                    agg[short(k)][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return agg


def trace_table(sub):
    agg, meta = defaultdict(list), {}
    for f in glob.glob(os.path.join(root, sub, "**", "*kernel_trace.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            k = short(r.get("Kernel_Name", "?"))
            agg[k].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
            meta[k] = (r.get("VGPR_Count"), r.get("SGPR_Count"), r.get("LDS_Block_Size"), r.get("Workgroup_Size"),
                       r.get("Grid_Size"))
    return agg, meta


if as_json:
    pmc = pmc_tables()
    durs, _ = trace_table("zoo_trace")
    out = {}
    for k, cs in pmc.items():
        if "FETCH_SIZE" not in cs or "WRITE_SIZE" not in cs:
            continue
        fetch = sum(cs["FETCH_SIZE"]) / len(cs["FETCH_SIZE"])
        write = sum(cs["WRITE_SIZE"]) / len(cs["WRITE_SIZE"])
        e = {"fetch_size_kib": fetch, "write_size_kib": write, "hbm_bytes_per_launch": (2 * fetch + write) * 1024,
             "dispatches": len(cs["FETCH_SIZE"]),
             "source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE over tools/prof_workload.py (FETCH_SIZE doubled for gfx950)"}
        if k in durs:
            e["avg_us_in_kernel_trace"] = sum(durs[k]) / len(durs[k])
        out[k] = e
    if "--stamp" in sys.argv:  # source hash of the library the passes ran on (tools/csrc_hash.py = cntt_version())
        out["csrc_hash"] = sys.argv[sys.argv.index("--stamp") + 1]
    print(json.dumps(out, indent=1, sort_keys=True))
    sys.exit(0)

for sub, title in (("trace", "bench.py (the driver's command)"), ("zoo_trace", "tools/prof_workload.py (kernel zoo)")):
    agg, meta = trace_table(sub)
    if not agg:
        continue
    print("# kernel trace of %s (durations in us)" % title)
    tot = sum(sum(v) for v in agg.values())
    for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
        print("%-96s calls=%4d avg=%9.2f min=%9.2f max=%9.2f total=%10.1f (%5.1f%%) vgpr/sgpr/lds/wg/grid=%s" % (
            k, len(v), sum(v) / len(v), min(v), max(v), sum(v), 100 * sum(v) / tot, meta[k]))
    for f in glob.glob(os.path.join(root, sub, "**", "*kernel_stats.csv"), recursive=True):
        print("# rocprofv3 --stats:", os.path.relpath(f, root))
        print(open(f).read()[:6000])

print("\n# PMC (per-dispatch average by kernel; tools/prof_workload.py)")
pmc = pmc_tables()
for k in sorted(pmc):
    cs = pmc[k]
    print(k)
    for c in sorted(cs):
        v = cs[c]
        print("    %-24s n=%3d avg=%16.1f" % (c, len(v), sum(v) / len(v)))
    if "FETCH_SIZE" in cs and "WRITE_SIZE" in cs:
        f, w = sum(cs["FETCH_SIZE"]) / len(cs["FETCH_SIZE"]), sum(cs["WRITE_SIZE"]) / len(cs["WRITE_SIZE"])
        print("    %-24s %.1f MiB  (2 x FETCH_SIZE + WRITE_SIZE)" % ("HBM bytes per launch", (2 * f + w) / 1024))
