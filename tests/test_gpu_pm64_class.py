"""CLS_PM64 (64-bit words, p = 2^64 - c with c < 2^32: the Solinas prime of src/prime64/generic_solinas.rs:35-40 and the
largest primes below 2^64): lazy representatives folded with 2^64 = c instead of Montgomery products.  Bit-exact parity
with the oracle on every LDS-resident size, on the extreme inputs of the carry analysis (all p-1, values next to 2^64 - c),
through the fused product and the fused chains, and against the same plan forced onto the Montgomery class."""
import os
import subprocess
import sys

import numpy as np
import pytest

import concrete_ntt_amd as cntt

from concrete_ntt_amd import prime64

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SOLINAS = 18446744069414584321          # 2^64 - 2^32 + 1: c = 2^32 - 1, the largest admissible c
P64 = 18446744073707716609              # benches/ntt.rs:117: largest prime = 1 mod 2^16 below 2^64 (c = 1835007)


def to_dev(a):
    import torch
    return torch.from_numpy(a.view(np.int64).copy()).cuda()


def to_host(t):
    return t.cpu().numpy().view(np.uint64)


def edge_polys(n, p, oracle, seed):
    rows = [np.full(n, p - 1, dtype=np.uint64), np.zeros(n, dtype=np.uint64)]
    alt = np.zeros(n, dtype=np.uint64)
    alt[::2] = p - 1
    rows.append(alt)
    near = np.full(n, p - 2, dtype=np.uint64)
    near[1::3] = 1
    near[2::3] = (1 << 32) - 1
    rows.append(near)
    hi = np.full(n, (p >> 32) << 32, dtype=np.uint64)   # high word set, low word clear
    hi[1::2] = (1 << 32) + 1
    rows.append(hi % np.uint64(p))
    rows.append(oracle.fill_uniform(n, p, seed, 64))
    rows.append(oracle.fill_uniform(n, p, seed + 1, 64))
    return np.concatenate(rows)


def check_transforms(oracle, n, p, seed):
    plan, ref = prime64.Plan.try_new(n, p), oracle.Plan.try_new(n, p, 64)
    assert plan is not None and ref is not None, (n, p)
    x = edge_polys(n, p, oracle, seed)
    for name in ("fwd", "inv"):
        d = to_dev(x)
        getattr(plan, name + "_batch")(d)
        got, want = to_host(d), x.copy()
        getattr(ref, name + "_batch")(want, 4)
        bad = np.nonzero(got != want)[0]
        assert bad.size == 0, "%s n=%d p=%d: %d mismatches, first at poly %d index %d" % (
            name, n, p, bad.size, bad[0] // n, bad[0] % n)
        assert int(got.max()) < p
    return plan


@pytest.mark.parametrize("p", [SOLINAS, P64])
@pytest.mark.parametrize("logn", list(range(4, 16)))
def test_pm64_every_size_vs_oracle(oracle, logn, p):
    n = 1 << logn
    plan = check_transforms(oracle, n, p, 6000 + logn)
    assert plan.info().arith_class == (5 if logn <= 14 else 2)


def test_pm64_class_boundary(oracle):
    """c just below 2^32 (Solinas has the largest c of all), a tiny c, and a 64-bit prime whose c needs 33 bits."""
    small_c = oracle.largest_prime_in_arithmetic_progression64(1 << 8, 1, (1 << 64) - (1 << 20), (1 << 64) - 1)
    assert check_transforms(oracle, 64, small_c, 5).info().arith_class == 5
    big_c = oracle.largest_prime_in_arithmetic_progression64(1 << 12, 1, 1 << 63, (1 << 64) - (1 << 33))
    assert (1 << 64) - big_c >= (1 << 32)
    assert prime64.Plan.try_new(1024, big_c).info().arith_class == 2


@pytest.mark.parametrize("p", [SOLINAS, P64])
@pytest.mark.parametrize("n", [16, 256, 1024, 2048, 4096])
def test_pm64_fused_product_equals_three_calls(oracle, n, p):
    plan, ref = prime64.Plan.try_new(n, p), oracle.Plan.try_new(n, p, 64)
    for batch in (1, 13, 301):
        a = oracle.fill_uniform(batch * n, p, 31 + batch, 64)
        b = oracle.fill_uniform(batch * n, p, 97 + batch, 64)
        a[:n] = p - 1
        b[:n] = p - 1
        want, bn = a.copy(), b.copy()
        ref.fwd_batch(bn, 4)
        ref.fwd_batch(want, 4)
        ref.mul_assign_normalize(want, bn)
        ref.inv_batch(want, 4)
        da = to_dev(a)
        plan.mul_ntt_batch(da, to_dev(bn))
        got = to_host(da)
        bad = np.nonzero(got != want)[0]
        assert bad.size == 0, "n=%d batch=%d: %d mismatches, first at %d" % (n, batch, bad.size, bad[0])


@pytest.mark.parametrize("n,J,O,batch", [(1024, 6, 2, 5), (2048, 9, 3, 2), (256, 25, 4, 7)])
@pytest.mark.parametrize("accumulate", [False, True])
@pytest.mark.parametrize("p", [SOLINAS, P64])
def test_pm64_mul_accumulate_chain(oracle, n, J, O, batch, accumulate, p):
    plan, ref = prime64.Plan.try_new(n, p), oracle.Plan.try_new(n, p, 64)
    terms = oracle.fill_uniform(batch * J * n, p, 11 + n, 64)
    key = oracle.fill_uniform(J * O * n, p, 22 + n, 64)
    init = oracle.fill_uniform(batch * O * n, p, 33 + n, 64)
    terms[: J * n] = p - 1
    key[:n] = p - 1
    want = init.copy() if accumulate else np.zeros(batch * O * n, dtype=np.uint64)
    tn = terms.copy()
    ref.fwd_batch(tn, 4)
    for b in range(batch):
        for o in range(O):
            acc = np.zeros(n, dtype=np.uint64)
            for j in range(J):
                ref.mul_accumulate(acc, tn[(b * J + j) * n:(b * J + j + 1) * n], key[(j * O + o) * n:(j * O + o + 1) * n])
            ref.inv(acc)
            sl = slice((b * O + o) * n, (b * O + o + 1) * n)
            want[sl] = (want[sl].astype(object) + acc.astype(object)) % p if accumulate else acc
    dout = to_dev(init if accumulate else np.zeros(batch * O * n, dtype=np.uint64))
    plan.external_product_batch(dout, to_dev(terms), to_dev(key), J, O, accumulate)
    assert np.array_equal(to_host(dout), want.astype(np.uint64))


@pytest.mark.parametrize("p", [SOLINAS, P64])
def test_pm64_equals_montgomery_class_on_a_large_batch(p):
    code = r'''
import hashlib
import torch
import concrete_ntt_amd as cntt
from concrete_ntt_amd import prime64
p, n, batch = %d, 1024, 4096
plan = prime64.Plan.try_new(n, p)
a = torch.empty(batch * n, dtype=torch.int64, device="cuda"); b = torch.empty_like(a)
cntt.fill_uniform(a, p, 5); cntt.fill_uniform(b, p, 6)
out = [plan.info().arith_class]
x = a.clone(); plan.fwd_batch(x); out.append(hashlib.sha256(x.cpu().numpy().tobytes()).hexdigest())
x = a.clone(); plan.inv_batch(x); out.append(hashlib.sha256(x.cpu().numpy().tobytes()).hexdigest())
x = a.clone(); plan.mul_ntt_batch(x, b); out.append(hashlib.sha256(x.cpu().numpy().tobytes()).hexdigest())
''' % p
    res = []
    for on in (1, 0):   # the switch is read when the plan is created (include/cntt.h, "testing only"): same process, two plans
        with cntt.debug_switches(pm64=on):
            ns = {}
            exec(code, ns)
        res.append([str(x) for x in ns["out"]])
    assert res[0][0] == "5" and res[1][0] == "2"
    assert res[0][1:] == res[1][1:]
