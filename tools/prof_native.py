#!/usr/bin/env python3
"""The two whole-product BASELINE shapes (C3 native64 N=4096 x 16384, C5 native_binary64 N=2048 x 65536) a few times,
for the rocprofv3 counter passes of tools/prof_native.sh.    python3 tools/prof_native.py [reps]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import concrete_ntt_amd as cntt  # noqa: E402
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import switches  # noqa: E402  (CNTT_SWITCHES="key=value,..." -> cntt_debug_set, tools/switches.py)
switches.apply()
from concrete_ntt_amd import native64, native_binary64  # noqa: E402

REPS = int(sys.argv[1]) if len(sys.argv) > 1 else 4
for cls, n, batch, binary in ((native64.Plan32, 4096, 16384, False), (native_binary64.Plan32, 2048, 65536, True)):
    plan = cls.try_new(n)
    lhs = torch.empty(batch * n, dtype=torch.int64, device="cuda")
    rhs = torch.empty_like(lhs)
    prod = torch.empty_like(lhs)
    cntt.fill_uniform(lhs, 0, 5)
    cntt.fill_uniform(rhs, 0, 6)
    if binary:
        rhs &= 1
    plan.reserve(batch)
    for _ in range(REPS):
        plan.negacyclic_polymul_batch(prod, lhs, rhs)
    torch.cuda.synchronize()
