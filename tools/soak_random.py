#!/usr/bin/env python3
"""Extended run of the seeded random GPU parity sweeps (tests/test_gpu_random.py) over many more seeds than the
test suite uses: prime plans, native polymul plans, product plans.  Prints the failing seeds, if any.
    python tools/soak_random.py [extra_seeds_per_family] [plans|native|product|chain|wrap]
`chain`: the fused mul_accumulate chain kernels with batches of more than two rounds of their persistent grids (random
primes of every class, random sizes / terms / outputs; tests/test_external_product_multitrip.py::run_chain_case)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle import pyoracle as oracle  # noqa: E402  (checker)

oracle.lib()
import test_gpu_random as t  # noqa: E402

extra = int(sys.argv[1]) if len(sys.argv) > 1 else 200
only = sys.argv[2] if len(sys.argv) > 2 else None
bad = 0
for name, fn, first in (("plans", t.test_gpu_random_plans, 24), ("native", t.test_gpu_random_native_polymul, 20),
                        ("product", t.test_gpu_random_product_plans, 12)):
    if only and only != name:
        continue
    for seed in range(first, first + extra):
        try:
            fn(oracle, seed)
        except BaseException as e:  # pytest.skip raises a BaseException subclass
            if type(e).__name__ == "Skipped":
                continue
            bad += 1
            print("FAIL", name, seed, repr(e)[:300], flush=True)
        if seed % 20 == 0:
            print(name, "seed", seed, flush=True)  # progress (a silent GPU job is taken for hung)
    print(name, "done", flush=True)
if only in (None, "chain"):
    import random
    import test_external_product_multitrip as mt
    for seed in range(extra if only == "chain" else max(extra // 10, 1)):
        rng = random.Random(7000 + seed)
        bits = 64 if seed % 2 == 0 else 32
        n = 1 << (rng.randint(10, 14) if bits == 64 else rng.randint(10, 12))   # the grid-round sizes of run_chain_case assume n >= 1024
        p = t._random_prime(oracle, rng, bits, n, top=seed % 3 == 0)
        J, O = rng.randint(1, 3), rng.randint(1, 2 if n == 16384 else 4)
        kernel = "blk" if bits == 64 and n >= 4096 else "wp"
        try:
            mt.run_chain_case(oracle, bits, n, p, J, O, kernel, bool(seed & 2), 100 + seed)
        except BaseException as e:
            if type(e).__name__ == "Skipped":
                continue
            bad += 1
            print("FAIL chain", seed, (bits, n, p, J, O), repr(e)[:300], flush=True)
        print("chain seed", seed, (bits, n, p, J, O), flush=True)
    print("chain done", flush=True)
if only in (None, "wrap"):
    # strict-class primes above 2^B / 3, where the reference's Barrett product can wrap: pointwise kernels, fused product, fused
    # chain against the oracle
    import random
    import test_gpu_parity as gp
    from concrete_ntt_amd import prime32, prime64
    for seed in range(extra if only == "wrap" else max(extra // 10, 1)):
        rng = random.Random(9000 + seed)
        bits = 64 if seed % 2 == 0 else 32
        n = 1 << rng.randint(4 if bits == 64 else 5, 12)
        hi = rng.randint((1 << bits) // 3, (1 << (bits - 1)) - 1)
        p = oracle.largest_prime_in_arithmetic_progression64(2 * n, 1, 0, hi)
        if p is None or p <= (1 << bits) // 3:
            continue
        plan, ref = (prime64 if bits == 64 else prime32).Plan.try_new(n, p), oracle.Plan.try_new(n, p, bits)
        if plan is None:
            continue
        try:
            nskip = gp.run_wrap_case(oracle, plan, ref, bits, n, p, max(2, 16384 // n) & ~1, False, seed)
            if nskip:
                print("wrap seed", seed, (bits, n, p), "chain outputs whose reference accumulators left the canonical range (compared like the rest):", nskip, flush=True)
        except BaseException as e:
            if type(e).__name__ == "Skipped":
                continue
            bad += 1
            print("FAIL wrap", seed, (bits, n, p), repr(e)[:300], flush=True)
        if seed % 20 == 0:
            print("wrap seed", seed, (bits, n, p), flush=True)
    print("wrap done", flush=True)
print("soak done, failures:", bad)
sys.exit(1 if bad else 0)
