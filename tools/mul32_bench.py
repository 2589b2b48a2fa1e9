#!/usr/bin/env python3
"""Fused product (mul_ntt_batch) against the three calls it replaces, 32-bit words at N = 8192 ... 32768.
    python tools/mul32_bench.py"""
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


def timed(fn, reps=10, ramp_s=0.5):
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < ramp_s:
        fn()
        torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


for name, p in (("lazy30", 1062862849), ("strict31", 2147352577), ("fpw32", 4293918721)):
    for n in (4096, 8192, 16384, 32768):
        plan = prime32.Plan.try_new(n, p)
        batch = (512 << 20) // (4 * n)
        a = torch.empty(batch * n, dtype=torch.int32, device="cuda")
        b = torch.empty_like(a)
        cntt.fill_uniform(a, p, 1)
        cntt.fill_uniform(b, p, 2)

        def three():
            plan.fwd_batch(a)
            plan.mul_assign_normalize_batch(a, b)
            plan.inv_batch(a)
        f = timed(lambda: plan.mul_ntt_batch(a, b))
        t = timed(three)
        print(json.dumps({"prime": name, "n": n, "batch": batch, "fused_ns": round(f * 1e6 / batch, 2),
                          "three_calls_ns": round(t * 1e6 / batch, 2)}), flush=True)
        del a, b, plan
        torch.cuda.empty_cache()
