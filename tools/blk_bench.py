#!/usr/bin/env python3
"""A/B timing of the large-N prime64 transforms (N = 4096 ... 32768) on HBM-resident batches.

One JSON line per (prime, n, direction): ns per transform, algorithmic bytes / time / 8 TB/s.  The kernel family is
chosen through the testing-only switchboard (CNTT_SWITCHES=blk=0 -> cntt_debug_set("blk", 0): one polynomial per workgroup, no persistent walk), which is read once per
process -- run the script once per setting:
    python tools/blk_bench.py --tag blk            > gpurun_out/blk.jsonl
    CNTT_SWITCHES=blk=0 python tools/blk_bench.py --tag onewg > gpurun_out/onewg.jsonl
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
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import switches  # noqa: E402  (CNTT_SWITCHES="key=value,..." -> cntt_debug_set, tools/switches.py)
switches.apply()
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
    ap.add_argument("--ext", default="", help="J,O: also time the fused mul_accumulate chain (external_product_batch) against the "
                                              "same step as separate fwd / mul_accumulate / inv calls")
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
            if args.ext:
                J, O = [int(x) for x in args.ext.split(",")]
                nb = max(1, batch // (2 * J))
                terms = torch.empty(nb * J * n, dtype=torch.int64, device="cuda")
                key = torch.empty(J * O * n, dtype=torch.int64, device="cuda")
                out = torch.zeros(nb * O * n, dtype=torch.int64, device="cuda")
                cntt.fill_uniform(terms, p, 3)
                cntt.fill_uniform(key, p, 4)
                # the same step as separate calls, on the layout that suits THEM best: term j of every element in one plane,
                # output o of every element in one plane, key[j][o] replicated per element (replication outside the timing)
                planes = torch.empty(J * nb * n, dtype=torch.int64, device="cuda")
                outp = torch.zeros(O * nb * n, dtype=torch.int64, device="cuda")
                keyrep = [[key.view(J, O, n)[j, o].repeat(nb).contiguous() for o in range(O)] for j in range(J)]
                tsrc = terms.view(nb, J, n).permute(1, 0, 2).contiguous().view(-1)

                def separate():   # what a caller of the reference writes: fwd every term, mul_accumulate, inv every output
                    planes.copy_(tsrc)
                    plan.fwd_batch(planes)
                    outp.zero_()
                    for j in range(J):
                        for o in range(O):
                            plan.mul_accumulate_batch(outp[o * nb * n:(o + 1) * nb * n], planes[j * nb * n:(j + 1) * nb * n], keyrep[j][o])
                    plan.inv_batch(outp)
                separate()
                sep_ms = round(timed(separate, max(2, args.reps // 3), args.ramp), 4)
                del keyrep, planes, tsrc
                ms = timed(lambda: plan.external_product_batch(out, terms, key, J, O, False), args.reps, args.ramp)
                print(json.dumps({"tag": args.tag, "prime": name, "n": n, "op": "ext_J%d_O%d" % (J, O), "batch": nb, "ms": round(ms, 4),
                                  "ns_per_element": round(ms * 1e6 / nb, 2), "separate_calls_ms": sep_ms,
                                  "hbm_frac": round((J + O) * n * 8 * nb / (ms * 1e-3) / 8e12, 4)}), flush=True)
                del terms, key, out, outp
            del a, plan
            torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
