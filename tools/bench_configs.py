#!/usr/bin/env python3
"""Secondary measurements (not the driver's bench line): the other BASELINE.json configs and the
prime32 path, at steady-state clocks, as JSON lines with roofline fractions on SURVEY 8(d) byte counts."""
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
from concrete_ntt_amd import native64, native_binary64, prime32, prime64  # noqa: E402

P62, P30 = 4611686018427322369, 1062862849
PEAK = 8000.0


def ramp(fn, seconds=1.5):
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < seconds:
        for _ in range(10):
            fn()
        torch.cuda.synchronize()


def timed(fn, reps):
    ramp(fn)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps  # ms


def emit(name, ms, units, unit, bytes_):
    gbs = bytes_ / (ms * 1e-3) / 1e9
    print(json.dumps({"case": name, "ms": round(ms, 4), "rate": units / (ms * 1e-3), "unit": unit,
                      "algorithmic_GBps": round(gbs, 1), "hbm_frac": round(gbs / PEAK, 4)}), flush=True)


def prime_case(mod, bits, n, p, batch, tag):
    plan = mod.Plan.try_new(n, p)
    dt = torch.int64 if bits == 64 else torch.int32
    a = torch.empty(batch * n, dtype=dt, device="cuda")
    cntt.fill_uniform(a, p, 1234)
    by = 2 * n * (bits // 8) * batch
    emit("%s fwd N=%d batch=%d" % (tag, n, batch), timed(lambda: plan.fwd_batch(a), 20), batch, "NTT/s", by)
    emit("%s inv N=%d batch=%d" % (tag, n, batch), timed(lambda: plan.inv_batch(a), 20), batch, "NTT/s", by)
    b = torch.empty_like(a)
    cntt.fill_uniform(b, p, 99)
    emit("%s mul_ntt (fwd+pointwise+inv) N=%d batch=%d" % (tag, n, batch), timed(lambda: plan.mul_ntt_batch(a, b), 10),
         2 * batch, "NTT/s", 2 * by)
    emit("%s mul_assign_normalize N=%d batch=%d" % (tag, n, batch), timed(lambda: plan.mul_assign_normalize_batch(a, b), 20),
         batch, "poly/s", 3 * n * (bits // 8) * batch)


def native_case(cls, n, batch, tag, binary):
    plan = cls.try_new(n)
    lhs = torch.empty(batch * n, dtype=torch.int64, device="cuda")
    rhs = torch.empty(batch * n, dtype=torch.int64, device="cuda")
    prod = torch.empty(batch * n, dtype=torch.int64, device="cuda")
    cntt.fill_uniform(lhs, 0, 5)
    cntt.fill_uniform(rhs, 0, 6)
    if binary:
        rhs &= 1
    plan.reserve(batch)
    ms = timed(lambda: plan.negacyclic_polymul_batch(prod, lhs, rhs), 5)
    emit("%s negacyclic_polymul N=%d batch=%d" % (tag, n, batch), ms, batch, "polymul/s", 3 * n * 8 * batch)


def product_case(n, primes, batch, tag):
    from concrete_ntt_amd import product
    big = 1
    for q in primes:
        big *= q
    plan = product.Plan.try_new(n, big, primes)
    std = torch.empty(batch * n, dtype=torch.int64, device="cuda")
    cntt.fill_uniform(std, big, 77)
    ntt = torch.zeros(batch * plan.ntt_domain_len(), dtype=torch.int64, device="cuda")
    rhs = torch.zeros_like(ntt)
    plan.fwd_batch(rhs, std)
    # algorithmic bytes (one pass): standard words + NTT-domain words
    by = (n + plan.ntt_domain_len()) * 8 * batch
    emit("%s fwd N=%d batch=%d" % (tag, n, batch), timed(lambda: plan.fwd_batch(ntt, std), 10), batch, "poly/s", by)
    emit("%s mul_assign_normalize N=%d batch=%d" % (tag, n, batch),
         timed(lambda: plan.mul_assign_normalize_batch(ntt, rhs), 10), batch, "poly/s", 3 * plan.ntt_domain_len() * 8 * batch)
    emit("%s inv(Replace) N=%d batch=%d" % (tag, n, batch),
         timed(lambda: plan.inv_batch(std, ntt, product.InvMode.Replace), 10), batch, "poly/s", by)
    emit("%s inv(Accumulate) N=%d batch=%d" % (tag, n, batch),
         timed(lambda: plan.inv_batch(std, ntt, product.InvMode.Accumulate), 10), batch, "poly/s", by + n * 8 * batch)


def ext_case(mod, bits, n, p, J, O, batch, tag):
    plan = mod.Plan.try_new(n, p)
    dt = torch.int64 if bits == 64 else torch.int32
    terms = torch.empty(batch * J * n, dtype=dt, device="cuda")
    key = torch.empty(J * O * n, dtype=dt, device="cuda")
    out = torch.zeros(batch * O * n, dtype=dt, device="cuda")
    cntt.fill_uniform(terms, p, 3)
    cntt.fill_uniform(key, p, 4)
    w = bits // 8
    ms = timed(lambda: plan.external_product_batch(out, terms, key, J, O), 10)
    emit("%s fused chain N=%d J=%d O=%d batch=%d" % (tag, n, J, O, batch), ms, (J + O) * batch, "NTT/s",
         (J + O) * n * w * batch)
    # the same work as separate calls: J fwd, J*O mul_accumulate, O inv per element (timed per kind, summed)
    t1 = terms[: batch * n].clone()
    acc = torch.zeros_like(t1)
    kk = torch.empty_like(t1)
    cntt.fill_uniform(kk, p, 5)
    f = timed(lambda: plan.fwd_batch(t1), 10)
    m = timed(lambda: plan.mul_accumulate_batch(acc, t1, kk), 10)
    i = timed(lambda: plan.inv_batch(t1), 10)
    sep = J * f + J * O * m + O * i
    emit("%s separate calls (J fwd + J*O mul_accumulate + O inv) N=%d J=%d O=%d batch=%d" % (tag, n, J, O, batch), sep,
         (J + O) * batch, "NTT/s", (2 * J + 3 * J * O + 2 * O) * n * w * batch)


def product_ext_case(n, primes, J, O, batch, tag):
    from concrete_ntt_amd import product
    big = 1
    for q in primes:
        big *= q
    plan = product.Plan.try_new(n, big, primes)
    terms = torch.empty(batch * J * n, dtype=torch.int64, device="cuda")
    cntt.fill_uniform(terms, 1 << 20, 3)
    g = torch.empty(J * O * n, dtype=torch.int64, device="cuda")
    cntt.fill_uniform(g, big, 4)
    key = torch.zeros(J * O * plan.ntt_domain_len(), dtype=torch.int64, device="cuda")
    plan.fwd_batch(key, g)
    out = torch.zeros(batch * O * n, dtype=torch.int64, device="cuda")
    for name, mode in (("Generic", product.FwdMode.Generic), ("Bounded", product.FwdMode.Bounded(1 << 20))):
        ms = timed(lambda: plan.external_product_batch(out, terms, key, J, O, mode, product.InvMode.Accumulate), 10)
        emit("%s external product (%s, Accumulate) N=%d J=%d O=%d batch=%d" % (tag, name, n, J, O, batch), ms, batch,
             "ext-products/s", (J + 2 * O) * n * 8 * batch)


if __name__ == "__main__":
    which = sys.argv[1:] or ["p64", "p32", "c4", "c3", "c5", "prod", "ext"]
    if "pext" in which:
        product_ext_case(2048, [4294955009, 4294914049], 6, 2, 4096, "product u32x2")
        product_ext_case(2048, [1073479681, 1062862849], 6, 2, 4096, "product u30x2")
        product_ext_case(2048, [18446744069414584321], 6, 2, 4096, "product u64x1 (Solinas)")
        product_ext_case(1024, [4611686018427322369], 6, 2, 8192, "product u64x1 (62-bit)")
    if "ext" in which:
        ext_case(prime64, 64, 1024, P62, 6, 2, 8192, "prime64 external product")
        ext_case(prime64, 64, 1024, P62, 12, 3, 4096, "prime64 external product")
        ext_case(prime32, 32, 2048, P30, 6, 2, 8192, "prime32 external product")
        ext_case(prime64, 64, 2048, P62, 6, 2, 4096, "prime64 external product")
        ext_case(prime64, 64, 2048, 18446744069414584321, 6, 2, 4096, "prime64 (Solinas) external product")
    if "prod" in which:
        product_case(2048, [4294955009, 4294914049], 32768, "product u32x2")
        product_case(2048, [18446744069414584321], 32768, "product u64x1 (Solinas)")
        product_case(1024, [61441, 59393, 40961, 18433], 65536, "product u32x4")
    if "p64" in which:
        prime_case(prime64, 64, 1024, P62, 65536, "prime64 (C2)")
    if "p32" in which:
        prime_case(prime32, 32, 1024, P30, 131072, "prime32")
        prime_case(prime32, 32, 4096, P30, 32768, "prime32")
    if "c4" in which:
        prime_case(prime64, 64, 16384, P62, 4096, "prime64 (C4 shard slice)")
        prime_case(prime64, 64, 4096, P62, 16384, "prime64")
    if "bign" in which:   # the persistent large-N kernels: u64 4096..16384, u32 8192..16384
        for n, b in ((4096, 16384), (8192, 8192), (16384, 4096)):
            prime_case(prime64, 64, n, P62, b, "prime64 62-bit")
        for n, b in ((8192, 16384), (16384, 8192), (32768, 4096)):
            prime_case(prime32, 32, n, P30, b, "prime32 30-bit")
    if "fp" in which:   # CLS_FP (p < 2^50): the 50-bit bench prime of benches/ntt.rs:112
        P50 = 1125899904679937
        prime_case(prime64, 64, 256, P50, 262144, "prime64 50-bit")
        prime_case(prime64, 64, 1024, P50, 65536, "prime64 50-bit")
        prime_case(prime64, 64, 2048, P50, 32768, "prime64 50-bit")
        prime_case(prime64, 64, 4096, P50, 16384, "prime64 50-bit")
        prime_case(prime64, 64, 16384, P50, 4096, "prime64 50-bit")
    if "fp51" in which:   # CLS_FP51: the 51-bit bench prime of benches/ntt.rs:113
        P51 = 2251799813554177
        for n, b in ((1024, 65536), (4096, 16384), (16384, 4096)):
            prime_case(prime64, 64, n, P51, b, "prime64 51-bit")
    if "pm64" in which:   # CLS_PM64: Solinas and the largest prime below 2^64 (benches/ntt.rs:116-117)
        for pp, tag in ((18446744069414584321, "Solinas"), (18446744073707716609, "64-bit")):
            for n, b in ((1024, 65536), (4096, 16384)):
                prime_case(prime64, 64, n, pp, b, "prime64 " + tag)
    if "p64n2048" in which:
        prime_case(prime64, 64, 2048, P62, 32768, "prime64")
    if "c3" in which:
        native_case(native64.Plan32, 4096, 16384, "native64::Plan32 (C3)", False)
    if "c5" in which:
        native_case(native_binary64.Plan32, 2048, 65536, "native_binary64::Plan32 (C5)", True)
