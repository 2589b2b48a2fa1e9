#!/usr/bin/env python3
"""A/B timing of the 32-bit-prime (p >= 2^31) paths: CLS_FPW against the Montgomery class (CNTT_SWITCHES=fp=0).
    python tools/fpw_bench.py ; CNTT_SWITCHES=fp=0 python tools/fpw_bench.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import concrete_ntt_amd as cntt  # noqa: E402
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import switches  # noqa: E402  (CNTT_SWITCHES="key=value,..." -> cntt_debug_set, tools/switches.py)
switches.apply()
from concrete_ntt_amd import prime32  # noqa: E402

P32 = 4293918721


def t(fn, reps=20):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


for n, batch in ((1024, 65536), (2048, 32768), (4096, 16384)):
    plan = prime32.Plan.try_new(n, P32)
    a = torch.empty(batch * n, dtype=torch.int32, device="cuda")
    b = torch.empty_like(a)
    cntt.fill_uniform(a, P32, 1)
    cntt.fill_uniform(b, P32, 2)
    J, O, eb = 6, 2, batch // 8
    out = torch.zeros(eb * O * n, dtype=torch.int32, device="cuda")
    print("class %d  N=%d  fwd %.3f ms  inv %.3f ms  fused product %.3f ms  chain J=6 O=2 x %d: %.3f ms" % (
        plan.info().arith_class, n, t(lambda: plan.fwd_batch(a)), t(lambda: plan.inv_batch(a)),
        t(lambda: plan.mul_ntt_batch(a, b)), eb,
        t(lambda: plan.external_product_batch(out, a[: eb * J * n], b[: J * O * n], J, O))), flush=True)
