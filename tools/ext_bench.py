#!/usr/bin/env python3
"""Fused mul_accumulate chain (external_product_batch), every modulus class, 1 ... 4 outputs: ms per call and ns per batch
element.  One JSON line per shape.
    python tools/ext_bench.py [--bits 32|64] [--tag T] [--sizes 2048,4096] [--outs 1,2,3,4]"""
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

PRIMES = {32: {"lazy30": 1073479681, "strict31": 2147352577, "fpw32": 4293918721},
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
    ap.add_argument("--sizes", default="2048,4096")
    ap.add_argument("--terms", type=int, default=6)
    ap.add_argument("--bits", type=int, default=32)
    ap.add_argument("--outs", default="1,2,3,4")
    args = ap.parse_args()
    J = args.terms
    mod, dt = (prime32, torch.int32) if args.bits == 32 else (prime64, torch.int64)
    for name, p in PRIMES[args.bits].items():
        for n in [int(x) for x in args.sizes.split(",")]:
            plan = mod.Plan.try_new(n, p)
            nb = ((32 << 20) // n) * 32 // args.bits     # 128 MiB per term plane
            terms = torch.empty(nb * J * n, dtype=dt, device="cuda")
            cntt.fill_uniform(terms, p, 3)
            for O in [int(x) for x in args.outs.split(",")]:
                key = torch.empty(J * O * n, dtype=dt, device="cuda")
                out = torch.zeros(nb * O * n, dtype=dt, device="cuda")
                cntt.fill_uniform(key, p, 4)
                ms = timed(lambda: plan.external_product_batch(out, terms, key, J, O, False))
                print(json.dumps({"tag": args.tag, "prime": name, "n": n, "J": J, "O": O, "batch": nb, "ms": round(ms, 4),
                                  "ns_per_element": round(ms * 1e6 / nb, 2)}), flush=True)
                del key, out
            del terms, plan
            torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
