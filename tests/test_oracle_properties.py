"""The reference's own property suite (it has no known-answer tests for NTT-domain values) run
against the oracle: it must hold for the oracle before the oracle may judge the GPU."""
import numpy as np
import pytest

U64_PRIMES = [1125899904679937, 2251799813554177, 4611686018427322369, 9223372036853661697,
              18446744069414584321, 18446744073707716609]
U32_PRIMES = [1062862849, 1073479681, 2147352577, 4293918721]


@pytest.mark.parametrize("bits,primes,sizes", [(64, U64_PRIMES, [16, 32, 64, 128, 256, 512, 1024]),
                                               (32, U32_PRIMES, [32, 64, 128, 256, 512, 1024])])
def test_product(oracle, bits, primes, sizes):
    """src/prime64.rs:1211-1267 / src/prime32.rs:1006-1060."""
    for p in primes:
        for n in sizes:
            plan = oracle.Plan.try_new(n, p, bits)
            lhs = oracle.fill_uniform(n, p, 11 * n + 1, bits)
            rhs = oracle.fill_uniform(n, p, 11 * n + 2, bits)
            conv = oracle.negacyclic_convolution(n, p, lhs, rhs, bits)
            fl, fr = lhs.copy(), rhs.copy()
            plan.fwd(fl)
            plan.fwd(fr)
            assert int(fl.max()) < p and int(fr.max()) < p
            # inv(fwd(a) . fwd(b)) == N * conv
            prod = np.array([(int(a) * int(b)) % p for a, b in zip(fl, fr)], dtype=fl.dtype)
            plan.inv(prod)
            assert int(prod.max()) < p
            assert prod.tolist() == [(int(c) * n) % p for c in conv]
            # inv(mul_assign_normalize(fwd a, fwd b)) == conv
            x = fl.copy()
            plan.mul_assign_normalize(x, fr)
            plan.inv(x)
            assert np.array_equal(x, conv)


def test_large_sizes_round_trip(oracle):
    """depth-first recursion above RECURSION_THRESHOLD (src/prime64.rs:7, src/prime32.rs:12)."""
    for bits, p, n in [(64, 4611686018427322369, 4096), (64, 18446744069414584321, 8192),
                       (32, 1062862849, 8192), (32, 4293918721, 4096)]:
        plan = oracle.Plan.try_new(n, p, bits)
        x = oracle.fill_uniform(n, p, 5, bits)
        y = x.copy()
        plan.fwd(y)
        plan.inv(y)
        plan.normalize(y)
        assert np.array_equal(x, y)


@pytest.mark.parametrize("kind", ["native32_plan32", "native64_plan32", "native_binary32_plan32",
                                  "native_binary64_plan32", "native32_plan52", "native64_plan52",
                                  "native_binary32_plan52", "native_binary64_plan52"])
def test_native_round_trip_and_polymul(oracle, kind):
    """src/native64.rs:1176-1243: inv(fwd(v)) == v * n (wrapping); polymul == schoolbook."""
    for n in (32, 64, 256):
        nat = oracle.Native(kind, n)
        dt = np.uint32 if nat.word == 4 else np.uint64
        wb = 32 if nat.word == 4 else 64
        v = (oracle.fill_uniform(n, 0, 17 + n, 64) >> np.uint64(64 - wb)).astype(dt)
        res = nat.residues()
        nat.fwd(v, res)
        out = np.zeros_like(v)
        nat.inv(out, res)
        assert np.array_equal(out, v * dt(n))
        rhs = (oracle.fill_uniform(n, 0, 99 + n, 64) >> np.uint64(64 - wb)).astype(dt)
        if nat.binary:
            rhs &= dt(1)
        prod = np.zeros_like(v)
        nat.negacyclic_polymul(prod, v, rhs)
        assert np.array_equal(prod, oracle.negacyclic_convolution(n, 0, v, rhs, wb))


# primes of the strict class (2^(B-2) <= p < 2^(B-1)) above 2^B / 3 on which the wrap below happens for about 7e-4 of uniform products
WRAP_PRIMES = [(32, 256, 2127586817), (64, 256, 8762203435012018177)]


def _reference_barrett(bits, p, a, b):
    """src/prime32.rs:398-401 / src/prime64.rs:549-552 on Python integers: the low B bits of d - c3 p."""
    big_q = p.bit_length()
    p_barrett = (1 << (big_q + bits - 1)) // p
    d = a * b
    c1 = (d >> (big_q - 1)) & ((1 << bits) - 1)
    c3 = (c1 * p_barrett) >> bits
    return (d - p * c3) & ((1 << bits) - 1), d - p * c3


@pytest.mark.parametrize("bits,n,p", WRAP_PRIMES)
def test_oracle_keeps_the_reference_barrett_wrap(oracle, bits, n, p):
    """The reference's Barrett product keeps the low B bits of d - c3 p, a remainder estimate in [0, 3p): for p > 2^B / 3 it can
    pass 2^B, and mul_assign_normalize / mul_accumulate are then off by 2^B mod p (src/prime32.rs:383-408, src/prime64.rs:534-584).
    The oracle restates that arithmetic, so it reproduces the wrap: checked against the formula on Python integers, and against the
    exact product to show that the wrap does happen on these inputs.  (Primes below 2^(B-2) -- 3p < 2^B -- cannot wrap.)"""
    plan = oracle.Plan.try_new(n, p, bits)
    dt = np.uint64 if bits == 64 else np.uint32
    count = 64 * n
    a, b = oracle.fill_uniform(count, p, 3, bits), oracle.fill_uniform(count, p, 4, bits)
    got = a.copy()
    for i in range(0, count, n):
        plan.mul_assign_normalize(got[i:i + n], b[i:i + n])
    acc0 = oracle.fill_uniform(count, p, 5, bits)
    acc = acc0.copy()
    for i in range(0, count, n):
        plan.mul_accumulate(acc[i:i + n], a[i:i + n], b[i:i + n])
    n_inv = pow(n, -1, p)
    n_inv_shoup = (n_inv << bits) // p
    mask, wraps = (1 << bits) - 1, 0
    for i in range(count):
        prod, full = _reference_barrett(bits, p, int(a[i]), int(b[i]))
        wraps += full > mask
        t = (prod * n_inv - ((prod * n_inv_shoup) >> bits) * p) & mask            # src/prime32.rs:403-406
        assert int(got[i]) == min(t, (t - p) & mask)
        prod = min(prod, (prod - p) & mask)                                       # src/prime64.rs:579-582
        s = (prod + int(acc0[i])) & mask
        assert int(acc[i]) == min(s, (s - p) & mask)
        exact = int(a[i]) * int(b[i]) * n_inv % p
        assert (int(got[i]) == exact) == (full <= mask)
    assert wraps > 0


def test_lazy_class_products_are_exact(oracle):
    """below 2^(B-2) the remainder estimate stays under 3p < 2^B: the reference's products are the exact ones (what lets the GPU's
    fused kernels use a Montgomery product for this class, csrc/ntt_arith.hpp mul_fused)."""
    for bits, n, p in [(32, 256, 1062862849), (64, 256, 4611686018427322369)]:
        plan = oracle.Plan.try_new(n, p, bits)
        a, b = oracle.fill_uniform(16 * n, p, 3, bits), oracle.fill_uniform(16 * n, p, 4, bits)
        got = a.copy()
        for i in range(0, 16 * n, n):
            plan.mul_assign_normalize(got[i:i + n], b[i:i + n])
        n_inv = pow(n, -1, p)
        assert all(int(got[i]) == int(a[i]) * int(b[i]) * n_inv % p for i in range(16 * n))
