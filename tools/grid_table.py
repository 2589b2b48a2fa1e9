#!/usr/bin/env python3
"""results.jsonl of tools/bench_grid.py -> the bench-id x N table kept under profiles/ (optionally next to an older grid).
    python tools/grid_table.py gpurun_out/r02_bench_grid/results.jsonl [profiles/r01_v6_bench_grid.jsonl] > table.txt"""
import json
import sys
from collections import OrderedDict


def load(path):
    rows = OrderedDict()
    for ln in open(path):
        ln = ln.strip()
        if not ln.startswith("{"):
            continue
        d = json.loads(ln)
        bid, _, n = d["bench_id"].rpartition("-")
        rows.setdefault(bid, {})[int(n)] = d
    return rows


cur = load(sys.argv[1])
old = load(sys.argv[2]) if len(sys.argv) > 2 else None
sizes = sorted({n for r in cur.values() for n in r})
print("tools/bench_grid.py on one MI355X: the reference's bench ids (benches/ntt.rs:84-235), batched device-resident launches "
      "with 1 GiB operands (HBM-resident: four times the 256 MiB Infinity Cache; the round-2 grid used 256 MiB).")
print("Cell = ns per call (one polynomial) / % of 8 TB/s on the algorithmic bytes (2*N*sizeof(T) per transform, 3*N*word per "
      "negacyclic_polymul)" + ("; second line = the same cell in %s." % sys.argv[2] if old else "."))
print()
print("%-30s" % "bench id \\ N" + "".join("%15d" % n for n in sizes))
for bid, r in cur.items():
    print("%-30s" % bid + "".join("%15s" % ("%.2f/%4.1f%%" % (r[n]["ns_per_call"], 100 * r[n]["hbm_frac"]) if n in r else "-")
                                  for n in sizes))
    if old and bid in old:
        o = old[bid]
        print("%-30s" % "   before" + "".join("%15s" % ("%.2f/%4.1f%%" % (o[n]["ns_per_call"], 100 * o[n]["hbm_frac"]) if n in o
                                                        else "-") for n in sizes))
