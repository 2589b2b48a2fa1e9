#!/usr/bin/env python3
"""The reference's own benchmark grid (benches/ntt.rs:84-235) on the MI355X batched path.

Same bench ids (`fwd-32-{p}-{n}`, `inv-64-{p}-{n}`, `native64-32-{n}`, `nativebinary128-32-{n}`, ...), same
polynomial sizes (256 ... 32768), same nine primes and the same native plans; for every id it writes the
reference's `benchmarks_parameters/<bench_id>/parameters.json` record (benches/ntt.rs:50-81) and one result line
    {"bench_id", "ns_per_call", "calls_per_s", "batch", "algorithmic_GBps", "hbm_frac"}
where a "call" is what criterion times in the reference (one fwd / inv / negacyclic_polymul of ONE polynomial);
here it is the batched device-resident launch divided by the batch.  Operands are 1 GiB each by default: four times the
256 MiB Infinity Cache (MALL), so every pass streams from HBM (round 2 used 256 MiB operands = exactly the MALL size, and
its HBM-leaning rows -- the double-precision classes, the pointwise kernels -- read up to 15 % high).

    python tools/bench_grid.py [--out gpurun_out/bench_grid] [--mib 1024] [--sizes 256,1024]
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
from concrete_ntt_amd import (native32, native64, native128, native_binary32, native_binary64,  # noqa: E402
                              native_binary128, prime32, prime64)

# largest_prime_in_arithmetic_progression64(1 << 16, 1, lo, hi) for the ranges of benches/ntt.rs:87-91, :111-118
P32 = [1073479681, 2147352577, 4293918721]
P64 = [1125899904679937, 2251799813554177, 4611686018427322369, 9223372036853661697,
       18446744069414584321, 18446744073707716609]
PEAK_GBPS = 8000.0


def prime_modulus(p):  # PrimeModulus::from_u64, benches/ntt.rs:27-48
    for bits, name in ((30, "FitsIn30Bits"), (31, "FitsIn31Bits"), (32, "FitsIn32Bits"), (50, "FitsIn50Bits"),
                       (51, "FitsIn51Bits"), (52, "FitsIn52Bits"), (62, "FitsIn62Bits"), (63, "FitsIn63Bits")):
        if p < 1 << bits:
            return name
    return "FitsIn64Bits"


def ramp(fn, seconds):
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < seconds:
        for _ in range(5):
            fn()
        torch.cuda.synchronize()


def timed(fn, reps, ramp_s):
    ramp(fn, ramp_s)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


class Grid:
    def __init__(self, out, mib, ramp_s):
        self.out, self.mib, self.ramp_s = out, mib, ramp_s
        os.makedirs(out, exist_ok=True)
        self.results = open(os.path.join(out, "results.jsonl"), "w")

    def batch_for(self, n, word):
        return max(64, (self.mib << 20) // (n * word))

    def record(self, bench_id, display, n, pm, p, ms, batch, bytes_per_call):
        d = os.path.join(self.out, "benchmarks_parameters", bench_id)
        os.makedirs(d, exist_ok=True)
        with open(os.path.join(d, "parameters.json"), "w") as f:
            json.dump({"display_name": display, "polynomial_size": n, "prime_modulus": pm, "prime_number": p}, f)
        ns = ms * 1e6 / batch
        gbs = bytes_per_call * batch / (ms * 1e-3) / 1e9
        line = {"bench_id": bench_id, "ns_per_call": round(ns, 3), "calls_per_s": round(1e9 / ns, 1), "batch": batch,
                "algorithmic_GBps": round(gbs, 1), "hbm_frac": round(gbs / PEAK_GBPS, 4)}
        self.results.write(json.dumps(line) + "\n")
        self.results.flush()
        print(json.dumps(line), flush=True)

    def prime(self, mod, bits, n, p):
        plan = mod.Plan.try_new(n, p)
        assert plan is not None, (bits, n, p)
        batch = self.batch_for(n, bits // 8)
        a = torch.empty(batch * n, dtype=torch.int64 if bits == 64 else torch.int32, device="cuda")
        cntt.fill_uniform(a, p, 0x5EED0000 + n)
        by = 2 * n * (bits // 8)
        for name, fn in (("fwd", plan.fwd_batch), ("inv", plan.inv_batch)):
            ms = timed(lambda: fn(a), 10, self.ramp_s)
            self.record("%s-%d-%d-%d" % (name, bits, p, n), "%s-%d" % (name, bits), n, prime_modulus(p), p, ms, batch, by)

    def native(self, cls, tag, pm, n, word, binary):
        plan = cls.try_new(n)
        assert plan is not None, (tag, n)
        batch = self.batch_for(n, word)
        words = batch * n * max(1, word // 8)
        lhs = torch.empty(words, dtype=torch.int64 if word >= 8 else torch.int32, device="cuda")
        rhs, prod = torch.empty_like(lhs), torch.empty_like(lhs)
        cntt.fill_uniform(lhs, 0, 5)
        if binary:
            rhs.zero_()
            low = torch.empty(batch * n, dtype=torch.int64, device="cuda")
            cntt.fill_uniform(low, 2, 6)
            if word == 16:
                rhs.view(-1, 2)[:, 0] = low
            else:
                rhs.copy_(low.to(rhs.dtype))
        else:
            cntt.fill_uniform(rhs, 0, 6)
        plan.reserve(batch)
        ms = timed(lambda: plan.negacyclic_polymul_batch(prod, lhs, rhs), 3, self.ramp_s)
        self.record("%s-%d" % (tag, n), tag, n, pm, 0, ms, batch, 3 * n * word)
        del plan, lhs, rhs, prod
        torch.cuda.empty_cache()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "bench_grid"))
    ap.add_argument("--mib", type=int, default=1024, help="MiB per operand (>= 512: past the 256 MiB Infinity Cache)")
    ap.add_argument("--ramp", type=float, default=0.25)
    ap.add_argument("--sizes", default="256,512,1024,2048,4096,8192,16384,32768")  # benches/ntt.rs:85
    ap.add_argument("--groups", default="p32,p64,n32,n64,n128")
    args = ap.parse_args()
    ns = [int(x) for x in args.sizes.split(",")]
    groups = args.groups.split(",")
    g = Grid(args.out, args.mib, args.ramp)
    warm = torch.empty(1 << 26, dtype=torch.int64, device="cuda")
    ramp(lambda: cntt.fill_uniform(warm, 0, 1), 2.0)  # DVFS ramp before the first timed case
    del warm
    for n in ns:
        if "p32" in groups:
            for p in P32:
                g.prime(prime32, 32, n, p)
        if "p64" in groups:
            for p in P64:
                g.prime(prime64, 64, n, p)
        if "n32" in groups:
            g.native(native32.Plan32, "native32-32", "Native32", n, 4, False)
            g.native(native_binary32.Plan32, "nativebinary32-32", "Native32", n, 4, True)
            g.native(native32.Plan52, "native32-52", "Native32", n, 4, False)
            g.native(native_binary32.Plan52, "nativebinary32-52", "Native32", n, 4, True)
        if "n64" in groups:
            g.native(native64.Plan32, "native64-32", "Native64", n, 8, False)
            g.native(native_binary64.Plan32, "nativebinary64-32", "Native64", n, 8, True)
            g.native(native64.Plan52, "native64-52", "Native64", n, 8, False)
            g.native(native_binary64.Plan52, "nativebinary64-52", "Native64", n, 8, True)
        if "n128" in groups:
            g.native(native128.Plan32, "native128-32", "Native128", n, 16, False)
            g.native(native_binary128.Plan32, "nativebinary128-32", "Native128", n, 16, True)


if __name__ == "__main__":
    main()
