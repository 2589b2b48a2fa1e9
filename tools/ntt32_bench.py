#!/usr/bin/env python3
"""Stand-alone 32-bit transforms at large N (HBM-resident batches): ns per transform and algorithmic bytes / time / 8 TB/s, forward
and inverse, one JSON line per (prime, n).  For same-box A/B runs of two builds: tools/ab_lib.sh <name> python tools/ntt32_bench.py
    python tools/ntt32_bench.py [--sizes 16384,32768] [--primes lazy30,strict31,fpw32] [--tag T]"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import concrete_ntt_amd as cntt  # noqa: E402
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import switches  # noqa: E402  (CNTT_SWITCHES="key=value,..." -> cntt_debug_set, tools/switches.py)
switches.apply()
from concrete_ntt_amd import prime32  # noqa: E402

PRIMES = {"lazy30": 1073479681, "strict31": 2147352577, "fpw32": 4293918721}


def timed(fn, reps=10, ramp_s=0.7):
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < ramp_s:
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--tag", default="")
    ap.add_argument("--sizes", default="16384,32768")
    ap.add_argument("--primes", default="lazy30,strict31,fpw32")
    ap.add_argument("--mib", type=int, default=1024)
    args = ap.parse_args()
    for name in args.primes.split(","):
        p = PRIMES[name]
        for n in [int(x) for x in args.sizes.split(",")]:
            plan = prime32.Plan.try_new(n, p)
            batch = (args.mib << 20) // (4 * n)
            a = torch.empty(batch * n, dtype=torch.int32, device="cuda")
            cntt.fill_uniform(a, p, 7)
            f = timed(lambda: plan.fwd_batch(a))
            i = timed(lambda: plan.inv_batch(a))
            cntt.fill_uniform(a, p, 7)   # (repeated transforms of the same buffer stay canonical; refill anyway)
            line = {"tag": args.tag, "prime": name, "n": n, "batch": batch}
            for d, ms in (("fwd", f), ("inv", i)):
                line[d + "_ns"] = round(ms * 1e6 / batch, 2)
                line[d + "_frac"] = round(2 * 4 * n * batch / (ms * 1e-3) / 8e12, 4)
            print(json.dumps(line), flush=True)
            del a, plan
            torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
