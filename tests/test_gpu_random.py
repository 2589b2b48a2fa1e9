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


def _random_prime(oracle, rng, bits, n, top=False):
    lo_bits = max(n.bit_length() + 2, 12)
    while True:
        nbits = rng.randint(bits - 2, bits) if top else rng.randint(lo_bits, bits)  # top: the three widest classes
        hi = rng.randint(1 << (nbits - 1), (1 << nbits) - 1)
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
    logn = rng.randint(4 if bits == 64 else 5, 13)
    n = 1 << logn
    p = _random_prime(oracle, rng, bits, n, top=seed % 3 == 0)
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
