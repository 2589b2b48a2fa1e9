#!/usr/bin/env python3
"""Fused product (mul_ntt_batch) by arithmetic class and size: ms per call, ns per product.  One JSON line per shape.
    python tools/mul_bench.py [--bits 32|64] [--sizes 1024,4096] [--tag T]"""
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
from concrete_ntt_amd import prime32, prime64  # noqa: E402

PRIMES = {32: {"lazy30": 1062862849, "strict31": 2147352577, "fpw32": 4293918721},
          64: {"lazy62": 4611686018427322369, "fp50": 1125899904679937, "strict63": 9223372036853661697,
               "solinas": 18446744069414584321}}


def timed(fn, reps=10, ramp_s=0.5):
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
    ap.add_argument("--sizes", default="1024,4096")
    ap.add_argument("--bits", type=int, default=64)
    args = ap.parse_args()
    mod, dt = (prime32, torch.int32) if args.bits == 32 else (prime64, torch.int64)
    for name, p in PRIMES[args.bits].items():
        for n in [int(x) for x in args.sizes.split(",")]:
            plan = mod.Plan.try_new(n, p)
            batch = (512 << 20) // (args.bits // 8 * n)
            a = torch.empty(batch * n, dtype=dt, device="cuda")
            b = torch.empty_like(a)
            cntt.fill_uniform(a, p, 1)
            cntt.fill_uniform(b, p, 2)
            ms = timed(lambda: plan.mul_ntt_batch(a, b))
            print(json.dumps({"tag": args.tag, "prime": name, "n": n, "batch": batch, "ms": round(ms, 4),
                              "ns_per_product": round(ms * 1e6 / batch, 2)}), flush=True)
            del a, b, plan
            torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
