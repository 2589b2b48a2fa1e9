#!/usr/bin/env python3
"""Timing of the product::Plan shape tfhe-rs uses (two 32-bit primes, src/product.rs:295-335 / :419-789): fused fwd and
inv (Replace / Accumulate) kernels, N = 2048, 32768 polynomials.  A/B against the Montgomery class:
    python tools/prod_bench.py ; CNTT_SWITCHES=fp=0 python tools/prod_bench.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, concrete_ntt_amd as cntt
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import switches  # noqa: E402  (CNTT_SWITCHES="key=value,..." -> cntt_debug_set, tools/switches.py)
switches.apply()
from concrete_ntt_amd import product
n, batch = 2048, 32768
primes = [4294955009, 4294914049]
big = primes[0] * primes[1]
plan = product.Plan.try_new(n, big, primes)
std = torch.empty(batch * n, dtype=torch.int64, device="cuda")
cntt.fill_uniform(std, big, 77)
ntt = torch.zeros(batch * plan.ntt_domain_len(), dtype=torch.int64, device="cuda")
def t(fn, reps=20):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
for _ in range(3):
    print("fwd %.3f ms  inv(replace) %.3f ms  inv(accumulate) %.3f ms" % (
        t(lambda: plan.fwd_batch(ntt, std)), t(lambda: plan.inv_batch(std, ntt, product.InvMode.Replace)),
        t(lambda: plan.inv_batch(std, ntt, product.InvMode.Accumulate))))
