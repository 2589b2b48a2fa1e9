#!/usr/bin/env python3
"""Counter-pass workload for the large-N prime64 transforms: fwd / inv (and the fused product) at N = 4096 ... 32768,
a few launches each, no ramps (tools/profile_one.sh runs it under rocprofv3).
    python3 tools/prof_large.py [reps] [sizes] [primes]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import concrete_ntt_amd as cntt  # noqa: E402
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import switches  # noqa: E402  (CNTT_SWITCHES="key=value,..." -> cntt_debug_set, tools/switches.py)
switches.apply()
from concrete_ntt_amd import prime64  # noqa: E402

PRIMES = {"lazy62": 4611686018427322369, "fp50": 1125899904679937, "strict63": 9223372036853661697,
          "solinas": 18446744069414584321}
REPS = int(sys.argv[1]) if len(sys.argv) > 1 else 4
SIZES = [int(x) for x in (sys.argv[2] if len(sys.argv) > 2 else "16384,8192,4096").split(",")]
NAMES = (sys.argv[3] if len(sys.argv) > 3 else "lazy62").split(",")

for name in NAMES:
    p = PRIMES[name]
    for n in SIZES:
        plan = prime64.Plan.try_new(n, p)
        batch = (512 << 20) // (8 * n)
        a = torch.empty(batch * n, dtype=torch.int64, device="cuda")
        b = torch.empty_like(a)
        cntt.fill_uniform(a, p, 1)
        cntt.fill_uniform(b, p, 2)
        for _ in range(REPS):
            plan.fwd_batch(a)
            plan.inv_batch(a)
            plan.mul_ntt_batch(a, b)
        torch.cuda.synchronize()
        del a, b, plan
