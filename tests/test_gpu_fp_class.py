"""CLS_FP / CLS_FP51 (64-bit words, p < 2^50 / p < 2^51: residues held as exact integers in doubles, v_fma_f64
butterflies -- the device counterparts of the reference's src/prime64/less_than_50bit.rs and less_than_51bit.rs
classes): bit-exact parity with the oracle's integer arithmetic on every LDS-resident size, at the edges of the magnitude bounds the kernel relies on (all-(p-1) inputs, the
largest admissible primes), through the fused product and the fused mul_accumulate chains, and against the same plan
forced onto the integer butterflies."""
import os
import subprocess
import sys

import numpy as np
import pytest

import concrete_ntt_amd as cntt

from concrete_ntt_amd import prime64

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
P50 = 1125899904679937            # benches/ntt.rs:112: largest prime = 1 mod 2^16 below 2^50
P51 = 2251799813554177            # benches/ntt.rs:113: ... below 2^51
PRIMES52 = [1125899881086977, 1125899885412353, 1125899886395393, 1125899899174913, 1125899902124033,
            1125899903107073]     # src/lib.rs:601-606


def to_dev(a):
    import torch
    return torch.from_numpy(a.view(np.int64).copy()).cuda()


def to_host(t):
    return t.cpu().numpy().view(np.uint64)


def edge_polys(n, p, oracle, seed):
    """Worst cases for the magnitude bounds: every coefficient p-1, alternating 0 / p-1, a spike, plus random rows."""
    rows = [np.full(n, p - 1, dtype=np.uint64), np.zeros(n, dtype=np.uint64)]
    alt = np.zeros(n, dtype=np.uint64)
    alt[::2] = p - 1
    rows.append(alt)
    half = np.full(n, p // 2, dtype=np.uint64)
    half[1::2] = p // 2 + 1
    rows.append(half)
    spike = np.zeros(n, dtype=np.uint64)
    spike[n - 1] = p - 1
    rows.append(spike)
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


@pytest.mark.parametrize("p,cls", [(P50, 3), (P51, 4)])
@pytest.mark.parametrize("logn", list(range(4, 16)))
def test_fp_every_size_vs_oracle(oracle, logn, p, cls):
    """N = 16 ... 16384 run in CLS_FP / CLS_FP51 (arith_class 3 / 4); N = 32768 falls back to the integer global-stage path."""
    n = 1 << logn
    plan = check_transforms(oracle, n, p, 4000 + logn)
    assert plan.info().arith_class == (cls if logn <= 14 else 0)


@pytest.mark.parametrize("p", PRIMES52 + [65537, 786433, "40-bit", "49-bit"])
def test_fp_other_primes(oracle, p):
    """The six Plan52 primes of the crate (src/lib.rs:601-606), small primes, a 40-bit and a 49-bit prime."""
    if isinstance(p, str):
        bits = int(p.split("-")[0])
        p = oracle.largest_prime_in_arithmetic_progression64(1 << 12, 1, 1 << (bits - 1), 1 << bits)
    for n in (16, 64, 1024, 2048):
        if (p - 1) % (2 * n):
            continue
        plan = check_transforms(oracle, n, p, p % 997 + n)
        assert plan.info().arith_class == 3


def test_fp_class_boundary(oracle):
    """The largest admissible primes of both classes (least headroom below 2^53), and the first class above."""
    lo = oracle.largest_prime_in_arithmetic_progression64(1 << 12, 1, 1 << 49, 1 << 50)
    assert (1 << 50) - lo < (1 << 24)
    for n in (64, 2048):
        assert check_transforms(oracle, n, lo, 77).info().arith_class == 3
    mid = oracle.largest_prime_in_arithmetic_progression64(1 << 12, 1, 1 << 50, 1 << 51)
    assert (1 << 51) - mid < (1 << 24)
    for n in (64, 2048):
        assert check_transforms(oracle, n, mid, 78).info().arith_class == 4
    hi = oracle.largest_prime_in_arithmetic_progression64(1 << 12, 1, 1 << 51, 1 << 52)
    assert prime64.Plan.try_new(1024, hi).info().arith_class == 0


@pytest.mark.parametrize("P50", [P50, P51])
@pytest.mark.parametrize("n", [16, 32, 256, 1024, 2048, 4096])
def test_fp_fused_product_equals_three_calls(oracle, n, P50):
    plan, ref = prime64.Plan.try_new(n, P50), oracle.Plan.try_new(n, P50, 64)
    for batch in (1, 13, 301):
        a = oracle.fill_uniform(batch * n, P50, 31 + batch, 64)
        b = oracle.fill_uniform(batch * n, P50, 97 + batch, 64)
        a[:n] = P50 - 1                      # extreme row
        b[:n] = P50 - 1
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
        if n <= 256 and batch == 1:
            assert np.array_equal(got, oracle.negacyclic_convolution(n, P50, a, b, 64))


@pytest.mark.parametrize("n,J,O,batch", [(1024, 6, 2, 5), (1024, 17, 1, 3), (2048, 9, 3, 2), (256, 25, 4, 7), (64, 8, 2, 9)])
@pytest.mark.parametrize("accumulate", [False, True])
@pytest.mark.parametrize("p", [P50, P51])
def test_fp_mul_accumulate_chain(oracle, n, J, O, batch, accumulate, p):
    """out[b][o] (+)= inv(sum_j fwd(terms[b][j]) . key[j][o]) in the fused chain kernel: more than eight terms exercise
    the accumulator's range reduction; the first element is the extreme one (every word p-1)."""
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


@pytest.mark.parametrize("P50,cls", [(P50, "3"), (P51, "4")])
def test_fp_equals_integer_butterflies_on_a_large_batch(P50, cls):
    """The same plan created under cntt_debug_set("fp", 0) (integer Shoup butterflies): identical bytes for fwd, inv
    and the fused product on 4096 random polynomials (a size the oracle would take long for is not needed: this is a
    device-vs-device check of two independent arithmetic paths)."""
    code = r'''
import hashlib, sys
import numpy as np, torch
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
''' % P50
    res = []
    for on in (1, 0):   # the switch is read when the plan is created (include/cntt.h, "testing only"): same process, two plans
        with cntt.debug_switches(fp=on):
            ns = {}
            exec(code, ns)
        res.append([str(x) for x in ns["out"]])
    assert res[0][0] == cls and res[1][0] == "0"
    assert res[0][1:] == res[1][1:]
