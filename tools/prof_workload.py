#!/usr/bin/env python3
"""Short kernel zoo for the rocprofv3 counter passes (tools/profile.sh): every hot kernel of the repo a few times at its
bench shape, no warm-up ramps (counters are per dispatch; durations come from the separate kernel-trace pass).

    python3 tools/prof_workload.py [reps]
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import concrete_ntt_amd as cntt  # noqa: E402
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import switches  # noqa: E402  (CNTT_SWITCHES="key=value,..." -> cntt_debug_set, tools/switches.py)
switches.apply()
from concrete_ntt_amd import native64, native_binary64, prime32, prime64, product  # noqa: E402

P62, P50, P51, P30 = 4611686018427322369, 1125899904679937, 2251799813554177, 1062862849
SOLINAS = 18446744069414584321
REPS = int(sys.argv[1]) if len(sys.argv) > 1 else 4


def prime(mod, bits, n, p, batch):
    plan = mod.Plan.try_new(n, p)
    dt = torch.int64 if bits == 64 else torch.int32
    a = torch.empty(batch * n, dtype=dt, device="cuda")
    b = torch.empty_like(a)
    cntt.fill_uniform(a, p, 1)
    cntt.fill_uniform(b, p, 2)
    for _ in range(REPS):
        plan.fwd_batch(a)
        plan.inv_batch(a)
        plan.mul_ntt_batch(a, b)
        plan.mul_assign_normalize_batch(a, b)
    torch.cuda.synchronize()


def native(cls, n, batch, binary):
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


def chain(n, p, J, O, batch):
    plan = prime64.Plan.try_new(n, p)
    terms = torch.empty(batch * J * n, dtype=torch.int64, device="cuda")
    key = torch.empty(J * O * n, dtype=torch.int64, device="cuda")
    out = torch.zeros(batch * O * n, dtype=torch.int64, device="cuda")
    cntt.fill_uniform(terms, p, 3)
    cntt.fill_uniform(key, p, 4)
    for _ in range(REPS):
        plan.external_product_batch(out, terms, key, J, O)
    torch.cuda.synchronize()


def prod_plan(n, primes, batch):
    big = 1
    for q in primes:
        big *= q
    plan = product.Plan.try_new(n, big, primes)
    std = torch.empty(batch * n, dtype=torch.int64, device="cuda")
    cntt.fill_uniform(std, big, 77)
    ntt = torch.zeros(batch * plan.ntt_domain_len(), dtype=torch.int64, device="cuda")
    for _ in range(REPS):
        plan.fwd_batch(ntt, std)
        plan.inv_batch(std, ntt, product.InvMode.Replace)
    torch.cuda.synchronize()


if __name__ == "__main__":
    prime(prime64, 64, 1024, P62, 65536)       # C2: mul_kernel_wp / ntt_kernel_wp / pointwise_kernel
    prime(prime64, 64, 16384, P62, 4096)       # C4 kernel: ntt_kernel_blk<u64, 14>
    prime(prime64, 64, 4096, P62, 16384)       # ntt_kernel_blk<u64, 12>, mul_kernel_blk
    prime(prime64, 64, 32768, P62, 2048)       # ntt_kernel_32k (single pass over HBM)
    prime(prime64, 64, 1024, P50, 65536)       # CLS_FP
    prime(prime64, 64, 1024, P51, 65536)       # CLS_FP51
    prime(prime64, 64, 1024, SOLINAS, 65536)   # generic class
    prime(prime32, 32, 1024, P30, 131072)
    prime(prime32, 32, 16384, P30, 8192)        # ntt_kernel (one polynomial per workgroup), mul_kernel_one
    native(native64.Plan32, 4096, 16384, False)          # C3: native_polymul_kernel_acc<1, 12, 256> (round 4: accumulating CRT, nothing parked)
    native(native_binary64.Plan32, 2048, 65536, True)    # C5: native_polymul_kernel_acc<4, 11, 256>
    native(native64.Plan32, 8192, 4096, False)           # native_polymul_kernel_acc<1, 13, 512> (one product per 512-thread workgroup)
    native(native64.Plan32, 16384, 2048, False)          # native_polymul_kernel_acc<1, 14, 1024>
    native(native64.Plan32, 32768, 1024, False)          # native_polymul_kernel_g<1, 15> (persistent, global parking, 32 coefficients per thread)
    chain(1024, P62, 6, 2, 8192)                          # ext_kernel_wp
    chain(4096, P62, 6, 2, 2048)                          # ext_kernel_blk
    chain(1024, P62, 6, 4, 4096)                          # ext_kernel_wp, four outputs (no next-term prefetch)
    chain(16384, P62, 6, 4, 512)                          # ext_kernel_blk<u64, 14, ..., 2> twice (four outputs as two launches)
    prod_plan(2048, [4294955009, 4294914049], 32768)      # product_fused
