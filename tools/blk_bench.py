#!/usr/bin/env python3
"""A/B timing of the large-N prime64 transforms (N = 4096 ... 32768) on HBM-resident batches.

One JSON line per (prime, n, direction): ns per transform, algorithmic bytes / time / 8 TB/s.  The kernel family is
chosen by the library from the environment (CNTT_DISABLE_BLK=1: one polynomial per workgroup, no persistent walk), which is read once per
process -- run the script once per setting:
    python tools/blk_bench.py --tag blk            > gpurun_out/blk.jsonl
    CNTT_DISABLE_BLK=1 python tools/blk_bench.py --tag onewg > gpurun_out/onewg.jsonl
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import concrete_ntt_amd as cntt  # noqa: E402
from concrete_ntt_amd import prime64  # noqa: E402

PRIMES = {"fp50": 1125899904679937, "fp51": 2251799813554177, "lazy62": 4611686018427322369,
          "strict63": 9223372036853661697, "solinas": 18446744069414584321, "pm64": 18446744073707716609}


def timed(fn, reps, ramp_s):
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
    ap.add_argument("--mib", type=int, default=1024)
    ap.add_argument("--sizes", default="4096,8192,16384")
    ap.add_argument("--primes", default="lazy62,fp50,strict63,solinas")
    ap.add_argument("--ramp", type=float, default=1.0)
    ap.add_argument("--reps", type=int, default=10)
    ap.add_argument("--mul", action="store_true", help="also time the fused fwd -> mul_assign_normalize -> inv call")
    args = ap.parse_args()
    for name in args.primes.split(","):
        p = PRIMES[name]
        for n in [int(x) for x in args.sizes.split(",")]:
            plan = prime64.Plan.try_new(n, p)
            batch = (args.mib << 20) // (8 * n)
            a = torch.empty(batch * n, dtype=torch.int64, device="cuda")
            cntt.fill_uniform(a, p, 0x5EED0000 + n)
            legs = [("fwd", lambda: plan.fwd_batch(a), 2), ("inv", lambda: plan.inv_batch(a), 2)]
            if args.mul:
                b = torch.empty_like(a)
                cntt.fill_uniform(b, p, 0x5EED1000 + n)
                legs.append(("mul_ntt", lambda: plan.mul_ntt_batch(a, b), 3))
            for leg, fn, words in legs:
                ms = timed(fn, args.reps, args.ramp)
                ns = ms * 1e6 / batch
                print(json.dumps({"tag": args.tag, "prime": name, "n": n, "op": leg, "batch": batch, "ms": round(ms, 4),
                                  "ns_per_poly": round(ns, 2),
                                  "hbm_frac": round(words * n * 8 * batch / (ms * 1e-3) / 8e12, 4)}), flush=True)
            del a, plan
            torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
