#!/usr/bin/env python3
"""Extended run of the seeded random GPU parity sweeps (tests/test_gpu_random.py) over many more seeds than the
test suite uses: prime plans, native polymul plans, product plans.  Prints the failing seeds, if any.
    python tools/soak_random.py [extra_seeds_per_family] [plans|native|product]"""
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
print("soak done, failures:", bad)
sys.exit(1 if bad else 0)
