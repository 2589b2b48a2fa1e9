"""Randomised parity sweep: random primes (every bit length the plans accept, found with the oracle's restatement
of prime::largest_prime_in_arithmetic_progression64), random sizes and ragged batches; fwd, inv,
mul_assign_normalize, the fused mul_ntt and the fused mul_accumulate chain against the oracle.  Seeds are fixed,
so the sweep is deterministic; bit-exact."""
import random

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _dev(a):
    import torch
    return torch.from_numpy(a.view(np.int64 if a.dtype == np.uint64 else np.int32).copy()).cuda()


def _host(t, dtype):
    return t.cpu().numpy().view(dtype)


def _random_prime(oracle, rng, bits, n, top=False, above_pow2=False):
    lo_bits = max(n.bit_length() + 2, 12)
    while True:
        nbits = rng.randint(bits - 2, bits) if top else rng.randint(lo_bits, bits)  # top: the three widest classes
        hi = rng.randint(1 << (nbits - 1), (1 << nbits) - 1)
        if above_pow2:
            # just ABOVE a power of two: where the reference's Barrett estimate reaches 2p and its mul_accumulate leaves words in [p, 2p)
            # (round 5, ADVICE round 4 -- primes drawn uniformly, or right below a power of two, never show it)
            hi = (1 << (nbits - 1)) + rng.randint(1, max(4 * n, (1 << (nbits - 1)) >> 6))
        p = oracle.largest_prime_in_arithmetic_progression64(2 * n, 1, 0, hi)
        if p is not None and p > 2 * n:
            return p


@pytest.mark.parametrize("seed", range(24))
def test_gpu_random_plans(oracle, seed):
    from concrete_ntt_amd import prime32, prime64
    rng = random.Random(1000 + seed)
    bits = 64 if seed % 2 == 0 else 32
    mod = prime64 if bits == 64 else prime32
    dt = np.uint64 if bits == 64 else np.uint32
    logn = rng.randint(4 if bits == 64 else 5, 15)
    n = 1 << logn
    p = _random_prime(oracle, rng, bits, n, top=seed % 3 == 0, above_pow2=seed % 5 == 4)
    plan, oplan = mod.Plan.try_new(n, p), oracle.Plan.try_new(n, p, bits)
    assert (plan is None) == (oplan is None), (n, p)
    if plan is None:
        return
    batch = rng.randint(1, 9)
    a = oracle.fill_uniform(batch * n, p, seed * 7 + 1, bits)
    b = oracle.fill_uniform(batch * n, p, seed * 7 + 2, bits)
    # forward / inverse
    fa = a.copy()
    for i in range(batch):
        oplan.fwd(fa[i * n:(i + 1) * n])
    d = _dev(a)
    plan.fwd_batch(d)
    assert np.array_equal(_host(d, dt), fa), ("fwd", n, p)
    ia = fa.copy()
    for i in range(batch):
        oplan.inv(ia[i * n:(i + 1) * n])
    plan.inv_batch(d)
    assert np.array_equal(_host(d, dt), ia), ("inv", n, p)
    # pointwise + fused product
    fb = b.copy()
    for i in range(batch):
        oplan.fwd(fb[i * n:(i + 1) * n])
    prod = fa.copy()
    oplan.mul_assign_normalize(prod, fb)
    want = prod.copy()
    for i in range(batch):
        oplan.inv(want[i * n:(i + 1) * n])
    d, dfb = _dev(a), _dev(fb)
    plan.mul_ntt_batch(d, dfb)
    assert np.array_equal(_host(d, dt), want), ("mul_ntt", n, p)
    dfa = _dev(fa)
    plan.mul_assign_normalize_batch(dfa, dfb)
    assert np.array_equal(_host(dfa, dt), prod), ("mul_assign_normalize", n, p)
    # fused mul_accumulate chain with J terms, O outputs taken from the same data
    J, O = rng.randint(1, 3), rng.randint(1, 4)
    eb = rng.randint(1, 3)
    terms = oracle.fill_uniform(eb * J * n, p, seed * 7 + 3, bits)
    key = oracle.fill_uniform(J * O * n, p, seed * 7 + 4, bits)
    exp = np.zeros(eb * O * n, dtype=dt)
    for e in range(eb):
        acc = [np.zeros(n, dtype=dt) for _ in range(O)]
        for j in range(J):
            t = terms[(e * J + j) * n:(e * J + j + 1) * n].copy()
            oplan.fwd(t)
            for o in range(O):
                oplan.mul_accumulate(acc[o], t, key[(j * O + o) * n:(j * O + o + 1) * n].copy())
        for o in range(O):
            oplan.inv(acc[o])
            exp[(e * O + o) * n:(e * O + o + 1) * n] = acc[o]
    dout = _dev(np.zeros(eb * O * n, dtype=dt))
    plan.external_product_batch(dout, _dev(terms), _dev(key), J, O)
    assert np.array_equal(_host(dout, dt), exp), ("external_product", n, p, J, O)


NATIVE_KINDS = ["native32_plan32", "native64_plan32", "native128_plan32", "native_binary32_plan32",
                "native_binary64_plan32", "native_binary128_plan32", "native32_plan52", "native64_plan52",
                "native_binary32_plan52", "native_binary64_plan52"]


@pytest.mark.parametrize("seed", range(20))
def test_gpu_random_native_polymul(oracle, seed):
    """Random native / native_binary plan, size and ragged batch; operands mix uniform words with the extremes
    (0, 1, all-ones) that stress the centred lift of the CRT; device-resident batched polymul vs the oracle, and
    polynomial 0 against the schoolbook wrapping convolution (the reference tests' own oracle)."""
    from concrete_ntt_amd import (native32, native64, native128, native_binary32, native_binary64, native_binary128)
    classes = {"native32_plan32": native32.Plan32, "native64_plan32": native64.Plan32, "native128_plan32": native128.Plan32,
               "native_binary32_plan32": native_binary32.Plan32, "native_binary64_plan32": native_binary64.Plan32,
               "native_binary128_plan32": native_binary128.Plan32, "native32_plan52": native32.Plan52,
               "native64_plan52": native64.Plan52, "native_binary32_plan52": native_binary32.Plan52,
               "native_binary64_plan52": native_binary64.Plan52}
    rng = random.Random(5000 + seed)
    kind = NATIVE_KINDS[seed % len(NATIVE_KINDS)]
    cls = classes[kind]
    n = 1 << rng.randint(5, 14)
    batch = rng.randint(1, 6)
    plan, ref = cls.try_new(n), oracle.Native(kind, n)
    assert plan is not None
    wpp = n * (2 if ref.word == 16 else 1)
    lhs = np.concatenate([ref.words() for _ in range(batch)])
    rhs = lhs.copy()
    raw = oracle.fill_uniform(lhs.size, 0, 31 * seed + 1, 64)
    raw2 = oracle.fill_uniform(lhs.size, 0, 31 * seed + 2, 64)
    lhs[:] = raw.astype(lhs.dtype) if lhs.dtype == np.uint64 else (raw >> np.uint64(32)).astype(np.uint32)
    rhs[:] = raw2.astype(rhs.dtype) if rhs.dtype == np.uint64 else (raw2 >> np.uint64(32)).astype(np.uint32)
    ones = lhs.dtype.type(np.iinfo(lhs.dtype).max)
    for k in range(0, lhs.size, 7):          # sprinkle extremes
        lhs[k] = (0, 1, ones)[k % 3]
    for k in range(3, rhs.size, 11):
        rhs[k] = (ones, 0, 1)[k % 3]
    if cls.BINARY:
        if ref.word == 16:
            rhs[0::2] &= np.uint64(1)
            rhs[1::2] = 0
        else:
            rhs &= rhs.dtype.type(1)
    want = np.zeros_like(lhs)
    ref.negacyclic_polymul_batch(want, lhs, rhs, batch, 2)
    dp = _dev(np.zeros_like(lhs))
    plan.negacyclic_polymul_batch(dp, _dev(lhs), _dev(rhs))
    got = _host(dp, lhs.dtype)
    assert np.array_equal(got, want), (kind, n, batch)
    if n <= 512:
        bits = ref.word * 8
        school = oracle.negacyclic_convolution(n, 0, lhs[:wpp].copy(), rhs[:wpp].copy(), 128) if bits == 128 else None
        if school is not None:
            assert np.array_equal(got[:wpp], school), (kind, n)


@pytest.mark.parametrize("seed", range(12))
def test_gpu_random_product_plans(oracle, seed):
    """Random product::Plan: 1-4 random distinct primes = 1 mod 2n whose product fits u64, random batch; fwd (Generic),
    inv (Replace / Accumulate) and the pointwise calls against the oracle, polynomial by polynomial."""
    import torch
    from concrete_ntt_amd import product
    rng = random.Random(9000 + seed)
    n = 1 << rng.randint(5, 11)
    k = rng.randint(1, 4)
    lp = oracle.largest_prime_in_arithmetic_progression64
    tries = 0
    while True:
        tries += 1
        if tries % 100 == 0 and k > 1:
            k -= 1  # too few NTT primes of that width for this size (e.g. four below 2^16 at n = 2048)
        bits_each = 64 // k
        primes = set()
        for _ in range(k):
            nb = rng.randint(max(n.bit_length() + 2, 12), bits_each)
            q = lp(2 * n, 1, 0, rng.randint(1 << (nb - 1), (1 << nb) - 1))
            if q:
                primes.add(q)
        big = 1
        for q in primes:
            big *= q
        if len(primes) == k and big < 2**64 and all(q >= 2**32 or n >= 32 for q in primes) and all(q < 2**32 or n >= 16 for q in primes):
            break
    primes = sorted(primes)
    plan, oplan = product.Plan.try_new(n, big, primes), oracle.Product.try_new(n, big, primes)
    assert (plan is None) == (oplan is None)
    if plan is None:
        return
    batch = rng.randint(1, 5)
    dl = plan.ntt_domain_len()
    n32 = sum(q < 2**32 for q in primes)
    n64 = k - n32
    std = oracle.fill_uniform(batch * n, big, seed + 1, 64)
    init = oracle.fill_uniform(batch * n, big, seed + 2, 64)

    def planes_to_poly(buf, i):
        parts = []
        if n32:
            w32 = buf[: (n // 2) * n32 * batch].view(np.uint32).reshape(n32, batch, n)
            parts.append(np.ascontiguousarray(w32[:, i, :]).reshape(-1).view(np.uint64))
        if n64:
            w64 = buf[(n // 2) * n32 * batch:].reshape(n64, batch, n)
            parts.append(np.ascontiguousarray(w64[:, i, :]).reshape(-1))
        return np.concatenate(parts)

    dstd = torch.from_numpy(std.view(np.int64).copy()).cuda()
    dntt = torch.zeros(dl * batch, dtype=torch.int64, device="cuda")
    plan.fwd_batch(dntt, dstd)
    hntt = dntt.cpu().numpy().view(np.uint64)
    refs = []
    for i in range(batch):
        r = np.zeros(dl, dtype=np.uint64)
        oplan.fwd(r, std[i * n:(i + 1) * n].copy())
        refs.append(r)
        assert np.array_equal(planes_to_poly(hntt, i), r), ("fwd", primes, n, i)
    sq = dntt.clone()
    plan.mul_assign_normalize_batch(sq, dntt)
    hsq = sq.cpu().numpy().view(np.uint64)
    for mode in (product.InvMode.Replace, product.InvMode.Accumulate):
        out = torch.from_numpy(init.view(np.int64).copy()).cuda()
        plan.inv_batch(out, sq.clone(), mode)
        hout = out.cpu().numpy().view(np.uint64)
        for i in range(batch):
            x = refs[i].copy()
            oplan.mul_assign_normalize(x, refs[i])
            assert np.array_equal(planes_to_poly(hsq, i), x), ("mul_assign_normalize", primes, n, i)
            r = init[i * n:(i + 1) * n].copy()
            oplan.inv(r, x, mode == product.InvMode.Accumulate)
            assert np.array_equal(hout[i * n:(i + 1) * n], r), ("inv", mode, primes, n, i)
