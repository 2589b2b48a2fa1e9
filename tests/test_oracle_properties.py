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
