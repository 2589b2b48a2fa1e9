"""CLS_FPW (32-bit words, p >= 2^31: the moduli with no lazy headroom in 32 bits): the LDS-resident transforms run on
doubles held in 64-bit register boxes, with centred int32 residues between the passes (csrc/ntt_arith.hpp).  Bit-exact
parity with the oracle's integer arithmetic on every LDS-resident size, on the worst cases of the magnitude analysis
(all-(p-1) inputs -- the inverse's sums double every stage --, the largest 32-bit primes), and against the same plan
forced onto the Montgomery class."""
import os
import subprocess
import sys

import numpy as np
import pytest

import concrete_ntt_amd as cntt

from concrete_ntt_amd import prime32

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
P32 = 4293918721                  # benches/ntt.rs:87-91: largest prime = 1 mod 2^16 below 2^32


def to_dev(a):
    import torch
    return torch.from_numpy(a.view(np.int32).copy()).cuda()


def to_host(t):
    return t.cpu().numpy().view(np.uint32)


def edge_polys(n, p, oracle, seed):
    rows = [np.full(n, p - 1, dtype=np.uint32), np.zeros(n, dtype=np.uint32)]
    alt = np.zeros(n, dtype=np.uint32)
    alt[::2] = p - 1
    rows.append(alt)
    half = np.full(n, p // 2, dtype=np.uint32)      # the two values either side of the centring threshold
    half[1::2] = p // 2 + 1
    rows.append(half)
    top = np.full(n, (1 << 31) - 1, dtype=np.uint32)  # either side of the int32 sign boundary
    top[1::2] = 1 << 31
    rows.append(np.minimum(top, np.uint32(p - 1)))
    spike = np.zeros(n, dtype=np.uint32)
    spike[n - 1] = p - 1
    rows.append(spike)
    rows.append(oracle.fill_uniform(n, p, seed, 32))
    rows.append(oracle.fill_uniform(n, p, seed + 1, 32))
    return np.concatenate(rows)


def check_transforms(oracle, n, p, seed):
    plan, ref = prime32.Plan.try_new(n, p), oracle.Plan.try_new(n, p, 32)
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
    # round trip through both directions: inv(fwd(x)) = n x
    d = to_dev(x)
    plan.fwd_batch(d)
    plan.inv_batch(d)
    plan.normalize_batch(d)
    assert np.array_equal(to_host(d), x)
    return plan


@pytest.mark.parametrize("logn", list(range(5, 17)))
def test_fpw_every_size_vs_oracle(oracle, logn):
    """N = 32 ... 32768 run in CLS_FPW (arith_class 6; round 4: N = 32768 too -- 11 / 20 spilled registers next to ordinary loads,
    still faster than the Montgomery class); the global-stage path of N = 65536 stays on the Montgomery class."""
    n = 1 << logn
    p = P32 if logn <= 15 else oracle.largest_prime_in_arithmetic_progression64(2 * n, 1, 1 << 31, 1 << 32)
    plan = check_transforms(oracle, n, p, 6000 + logn)
    assert plan.info().arith_class == (6 if logn <= 15 else 2)


def test_fpw_class_boundary(oracle):
    """The largest 32-bit primes (least headroom, centred residues next to +-2^31), the smallest prime above 2^31, and the
    largest below it (strict class, unchanged)."""
    for n in (64, 1024, 4096):
        hi = oracle.largest_prime_in_arithmetic_progression64(2 * n, 1, 1 << 31, 1 << 32)
        assert (1 << 32) - hi < (1 << 20), hi
        assert check_transforms(oracle, n, hi, 91).info().arith_class == 6
        lo = None
        k = ((1 << 31) // (2 * n)) + 1
        while lo is None:   # smallest prime = 1 mod 2n above 2^31 (a handful of candidates)
            c = k * 2 * n + 1
            if oracle.largest_prime_in_arithmetic_progression64(2 * n, 1, c, c + 1) == c:
                lo = c
            k += 1
        assert check_transforms(oracle, n, lo, 92).info().arith_class == 6
        below = oracle.largest_prime_in_arithmetic_progression64(2 * n, 1, 1 << 30, 1 << 31)
        assert prime32.Plan.try_new(n, below).info().arith_class == 1


@pytest.mark.parametrize("n", [32, 256, 1024, 4096])
def test_fpw_fused_product_and_chain(oracle, n):
    """The fused product and the fused mul_accumulate chain in CLS_FPW (pointwise steps on centred int32 patterns):
    same values as the oracle and as the three separate calls; the first polynomial is the extreme one (every word p-1)."""
    plan, ref = prime32.Plan.try_new(n, P32), oracle.Plan.try_new(n, P32, 32)
    batch = 9
    a = oracle.fill_uniform(batch * n, P32, 51, 32)
    b = oracle.fill_uniform(batch * n, P32, 52, 32)
    a[:n] = P32 - 1
    b[:n] = P32 - 1
    want, bn = a.copy(), b.copy()
    ref.fwd_batch(bn, 4)
    ref.fwd_batch(want, 4)
    ref.mul_assign_normalize(want, bn)
    ref.inv_batch(want, 4)
    da = to_dev(a)
    plan.mul_ntt_batch(da, to_dev(bn))
    assert np.array_equal(to_host(da), want)
    dc = to_dev(a)                       # the same through three separate calls
    plan.fwd_batch(dc)
    plan.mul_assign_normalize_batch(dc, to_dev(bn))
    plan.inv_batch(dc)
    assert np.array_equal(to_host(dc), want)
    J, O = 11, 3
    terms = oracle.fill_uniform(batch * J * n, P32, 61, 32)
    key = oracle.fill_uniform(J * O * n, P32, 62, 32)
    terms[: J * n] = P32 - 1
    key[:n] = P32 - 1
    exp = np.zeros(batch * O * n, dtype=np.uint32)
    tn = terms.copy()
    ref.fwd_batch(tn, 4)
    for e in range(batch):
        for o in range(O):
            acc = np.zeros(n, dtype=np.uint32)
            for j in range(J):
                ref.mul_accumulate(acc, tn[(e * J + j) * n:(e * J + j + 1) * n], key[(j * O + o) * n:(j * O + o + 1) * n])
            ref.inv(acc)
            exp[(e * O + o) * n:(e * O + o + 1) * n] = acc
    dout = to_dev(np.zeros(batch * O * n, dtype=np.uint32))
    plan.external_product_batch(dout, to_dev(terms), to_dev(key), J, O)
    assert np.array_equal(to_host(dout), exp)


def test_fpw_equals_montgomery_butterflies_on_a_large_batch():
    """The same plan created under cntt_debug_set("fp", 0) (Montgomery class): identical bytes for fwd and inv on 8192
    random polynomials of N = 1024 (plus the fused product and a fused chain) and 512 of N = 16384 -- a device-vs-device
    check of two independent arithmetic paths."""
    code = r'''
import hashlib
import torch
import concrete_ntt_amd as cntt
from concrete_ntt_amd import prime32
out = []
for n, batch in ((1024, 8192), (16384, 512)):
    p = %d
    plan = prime32.Plan.try_new(n, p)
    a = torch.empty(batch * n, dtype=torch.int32, device="cuda")
    cntt.fill_uniform(a, p, 5)
    out.append(plan.info().arith_class)
    x = a.clone(); plan.fwd_batch(x); out.append(hashlib.sha256(x.cpu().numpy().tobytes()).hexdigest())
    x = a.clone(); plan.inv_batch(x); out.append(hashlib.sha256(x.cpu().numpy().tobytes()).hexdigest())
    if n == 1024:
        b = torch.empty_like(a); cntt.fill_uniform(b, p, 6)
        x = a.clone(); plan.mul_ntt_batch(x, b); out.append(hashlib.sha256(x.cpu().numpy().tobytes()).hexdigest())
        o = torch.zeros(64 * 2 * n, dtype=torch.int32, device="cuda")
        plan.external_product_batch(o, a[: 64 * 5 * n], b[: 5 * 2 * n], 5, 2)
        out.append(hashlib.sha256(o.cpu().numpy().tobytes()).hexdigest())
''' % P32
    res = []
    for on in (1, 0):   # the switch is read when the plan is created (include/cntt.h, "testing only"): same process, two plans
        with cntt.debug_switches(fp=on):
            ns = {}
            exec(code, ns)
        res.append([str(x) for x in ns["out"]])
    # per size: class, sha(fwd), sha(inv); N = 1024 also sha(fused product), sha(fused chain)
    assert res[0][0] == "6" and res[0][5] == "6" and res[1][0] == "2" and res[1][5] == "2"
    assert len(res[0]) == 8 and [x for i, x in enumerate(res[0]) if i not in (0, 5)] == [x for i, x in enumerate(res[1]) if i not in (0, 5)]
