"""GPU parity tests: the HIP path (through the C ABI of include/cntt.h) against the CPU oracle, the
committed golden vectors, and size-independent properties at the BASELINE.json sizes.
Everything here is bit-exact (unsigned integer arithmetic; no tolerance)."""
import hashlib

import numpy as np
import pytest

import concrete_ntt_amd as cntt
from concrete_ntt_amd import (native32, native64, native128, native_binary32, native_binary64, native_binary128,
                              prime32, prime64)

pytestmark = pytest.mark.gpu

P62 = 4611686018427322369      # headline prime, 62-bit class (benches/ntt.rs:115)
U64_PRIMES = [1125899904679937, 2251799813554177, P62, 9223372036853661697, 18446744069414584321,
              18446744073707716609]                     # benches/ntt.rs:111-118: every dispatch class
U32_PRIMES = [1062862849, 1073479681, 2147352577, 4293918721]  # README + benches/ntt.rs:87-91


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def _mod(bits):
    return prime64 if bits == 64 else prime32


def _torch():
    import torch
    assert torch.cuda.is_available(), "these tests need the MI355X"
    return torch


def to_dev(a):
    torch = _torch()
    sdt = np.int64 if a.dtype == np.uint64 else np.int32
    return torch.from_numpy(a.view(sdt).copy()).cuda()


def to_host(t, dtype):
    return t.cpu().numpy().view(dtype)


@pytest.fixture(scope="module")
def plans():
    cache = {}

    def get(bits, n, p):
        key = (bits, n, p)
        if key not in cache:
            cache[key] = _mod(bits).Plan.try_new(n, p)
            assert cache[key] is not None, key
        return cache[key]
    return get


@pytest.fixture(scope="module")
def oplans(oracle):
    cache = {}

    def get(bits, n, p):
        key = (bits, n, p)
        if key not in cache:
            cache[key] = oracle.Plan.try_new(n, p, bits)
        return cache[key]
    return get


# ------------------------------------------------------------------------------------------------
def test_gpu_present_and_library_loaded():
    assert cntt.device_count() >= 1
    assert "gfx950" in cntt.version()


def test_readme_example(plans):
    """BASELINE.json configs[0] / README.md:41-57 through the host-slice API (C1)."""
    plan = plans(32, 32, 1062862849)
    data = np.arange(32, dtype=np.uint32)
    fwd = data.copy()
    plan.fwd(fwd)
    assert fwd[:8].tolist() == [8337849, 878691898, 914453352, 923715776, 1012328021, 392768238, 897146226,
                                61013893]
    inv = fwd.copy()
    plan.inv(inv)
    assert inv.tolist() == [32 * i for i in range(32)]


def test_transforms_match_golden(golden, oracle, plans):
    for ent in golden["transforms"]:
        bits, n, p = ent["bits"], ent["n"], ent["p"]
        plan = plans(bits, n, p)
        dt = plan.dtype
        if ent["input"] == "iota":
            x = (np.arange(n, dtype=np.uint64) % np.uint64(p)).astype(dt)
        else:
            x = oracle.fill_uniform(n, p, ent["seed"], bits)
        f = x.copy()
        plan.fwd(f)
        assert f[:8].tolist() == ent["fwd_head"], (bits, n, p, ent["input"])
        assert sha(f) == ent["fwd_sha256"], (bits, n, p, ent["input"])
        i = x.copy()
        plan.inv(i)
        assert i[:8].tolist() == ent["inv_head"], (bits, n, p, ent["input"])
        assert sha(i) == ent["inv_sha256"], (bits, n, p, ent["input"])


def _batch_case(oracle, plans, oplans, bits, n, p, batch, seed):
    plan, ref = plans(bits, n, p), oplans(bits, n, p)
    x = oracle.fill_uniform(batch * n, p, seed, bits)
    for name in ("fwd", "inv"):
        d = to_dev(x)
        getattr(plan, name + "_batch")(d)
        got = to_host(d, plan.dtype)
        want = x.copy()
        getattr(ref, name + "_batch")(want, 4)
        bad = np.nonzero(got != want)[0]
        assert bad.size == 0, "%s bits=%d n=%d p=%d: %d mismatches, first at %d (poly %d, index %d)" % (
            name, bits, n, p, bad.size, bad[0], bad[0] // n, bad[0] % n)
        assert int(got.max()) < p


@pytest.mark.parametrize("logn", list(range(4, 17)))
def test_u64_every_size_vs_oracle(oracle, plans, oplans, logn):
    """All N from the try_new minimum (16) through the LDS-resident sizes (<= 16384) and the
    global-stage path beyond; batch is ragged with respect to the polynomials-per-workgroup."""
    n = 1 << logn
    batch = 37 if n <= 4096 else 5
    # the headline prime is 1 mod 2^16 only (N <= 32768); beyond that use the largest 62-bit prime = 1 mod 2^18
    p = P62 if logn <= 15 else 4611686018425815041
    _batch_case(oracle, plans, oplans, 64, n, p, batch, 1000 + logn)


@pytest.mark.parametrize("logn", list(range(5, 17)))
def test_u32_every_size_vs_oracle(oracle, plans, oplans, logn):
    n = 1 << logn
    batch = 37 if n <= 4096 else 5
    _batch_case(oracle, plans, oplans, 32, n, 1062862849, batch, 2000 + logn)


@pytest.mark.parametrize("n,p", [(16384, 1062862849), (32768, 1062862849), (16384, 2147352577), (32768, 2147352577),
                                 (16384, 4293918721), (32768, 4293918721)])
def test_u32_wave_block_walk_takes_several_trips(oracle, plans, oplans, n, p):
    """32-bit words at n = 16384 / 32768 on the persistent wave-block walk (ntt_kernel_blk<u32>: two 512-thread / one 1024-thread
    workgroup per CU; round 4: p >= 2^31 and the 31-bit class's inverse at n = 32768 too, with ordinary loads instead of the asynchronous
    prefetch): two rounds of the resident grid plus a ragged third, every polynomial against the oracle, both directions."""
    import torch
    plan, ref = plans(32, n, p), oplans(32, n, p)
    per_round = torch.cuda.get_device_properties(0).multi_processor_count * (2 if n == 16384 else 1)
    batch = 2 * per_round + per_round // 3 + 3
    a = oracle.fill_uniform(batch * n, p, 4242 + n, 32)
    want = a.copy()
    ref.fwd_batch(want, 8)
    d = to_dev(a)
    plan.fwd_batch(d)
    got = to_host(d, plan.dtype)
    bad = np.nonzero((got != want).reshape(batch, n).any(axis=1))[0]
    assert bad.size == 0, ("fwd: polynomials differing from the oracle", bad[:8], batch, per_round)
    ref.inv_batch(want, 8)
    plan.inv_batch(d)
    got = to_host(d, plan.dtype)
    bad = np.nonzero((got != want).reshape(batch, n).any(axis=1))[0]
    assert bad.size == 0, ("inv: polynomials differing from the oracle", bad[:8], batch, per_round)


@pytest.mark.parametrize("p", U64_PRIMES)
@pytest.mark.parametrize("n", [16, 256, 1024, 4096])
def test_u64_every_class_vs_oracle(oracle, plans, oplans, p, n):
    _batch_case(oracle, plans, oplans, 64, n, p, 9, p % 1000 + n)


@pytest.mark.parametrize("p", U64_PRIMES)
@pytest.mark.parametrize("n", [2048, 8192, 16384, 32768])
def test_u64_every_class_large_sizes_vs_oracle(oracle, plans, oplans, p, n):
    """Every arithmetic class on the kernels of the larger sizes: the persistent L2-twiddle kernel (N = 4096 .. 16384 where
    it is used), one polynomial per workgroup, and the global-stage path (N = 32768); ragged batch of 5."""
    _batch_case(oracle, plans, oplans, 64, n, p, 5, p % 1000 + n)


@pytest.mark.parametrize("p", U32_PRIMES)
@pytest.mark.parametrize("n", [32, 256, 2048, 8192])
def test_u32_every_class_vs_oracle(oracle, plans, oplans, p, n):
    _batch_case(oracle, plans, oplans, 32, n, p, 9, p % 1000 + n)


def test_boundary_inputs(plans, oplans):
    """All-zero, all-(p-1) and single-spike polynomials (range edges of the lazy butterflies)."""
    for bits, n, p in [(64, 1024, P62), (64, 1024, 9223372036853661697), (64, 64, 18446744073707716609),
                       (32, 1024, 1073479681), (32, 1024, 2147352577), (32, 64, 4293918721)]:
        plan, ref = plans(bits, n, p), oplans(bits, n, p)
        dt = plan.dtype
        cases = [np.zeros(n, dtype=dt), np.full(n, p - 1, dtype=dt)]
        spike = np.zeros(n, dtype=dt)
        spike[n - 1] = p - 1
        cases.append(spike)
        for x in cases:
            for name in ("fwd", "inv"):
                got, want = x.copy(), x.copy()
                getattr(plan, name)(got)
                getattr(ref, name)(want)
                assert np.array_equal(got, want), (bits, n, p, name)


def test_words_outside_the_input_contract_do_not_fault(plans, oplans):
    """The documented behaviour for coefficients >= modulus (include/cntt.h): the transforms neither reject nor
    reduce them -- the lazy classes skip the first stage's conditional subtraction on the strength of the contract -- so the
    polynomial they sit in comes out as unspecified residues; the call must complete, and the OTHER polynomials of the batch
    must be exactly what they are without the offender (no state leaks between polynomials)."""
    import torch
    for bits, n, p in ((64, 1024, P62), (32, 1024, 1062862849), (64, 4096, P62), (32, 8192, 1062862849)):
        plan, oplan = plans(bits, n, p), oplans(bits, n, p)
        dt, tdt = (np.uint64, torch.int64) if bits == 64 else (np.uint32, torch.int32)
        batch = 5
        a = np.array([(i * 0x9E3779B97F4A7C15) % p for i in range(batch * n)], dtype=dt)
        bad = a.copy()
        top = (1 << bits) - 1
        bad[2 * n:3 * n] = np.array([p, p + 1, 2 * p if 2 * p <= top else top, top] * (n // 4), dtype=dt)   # polynomial 2: all outside
        want = a.copy()
        for i in range(batch):
            oplan.fwd(want[i * n:(i + 1) * n])
        d = torch.from_numpy(bad.view(np.int64 if bits == 64 else np.int32).copy()).cuda()
        plan.fwd_batch(d)      # completes
        got = d.cpu().numpy().view(dt)
        for i in (0, 1, 3, 4):
            assert np.array_equal(got[i * n:(i + 1) * n], want[i * n:(i + 1) * n]), (bits, n, i)
        plan.inv_batch(d)      # and the inverse of whatever came out completes too
        torch.cuda.synchronize()
        with pytest.raises(Exception):
            plan.check_canonical(bad)
        plan.check_canonical(a)


def test_pointwise_golden_and_oracle(golden, oracle, plans, oplans):
    for ent in golden["pointwise"]:
        bits, n, p, seed = ent["bits"], ent["n"], ent["p"], ent["seed"]
        plan = plans(bits, n, p)
        a, b, c = (oracle.fill_uniform(n, p, seed + k, bits) for k in range(3))
        x = a.copy()
        plan.mul_assign_normalize(x, b)
        assert x.tolist() == ent["mul_assign_normalize"], (bits, p)
        x = a.copy()
        plan.normalize(x)
        assert x.tolist() == ent["normalize"], (bits, p)
        acc = c.copy()
        plan.mul_accumulate(acc, a, b)
        assert acc.tolist() == ent["mul_accumulate"], (bits, p)
    # batched, device-resident, odd element counts exercise the vector tail
    for bits, n, p in [(64, 1024, P62), (64, 16, 18446744069414584321), (32, 2048, 1062862849),
                       (32, 32, 4293918721), (64, 256, 9223372036853661697), (32, 64, 2147352577)]:
        plan, ref = plans(bits, n, p), oplans(bits, n, p)
        batch = 7
        a, b, c = (oracle.fill_uniform(batch * n, p, 77 + k, bits) for k in range(3))
        da, db, dc = to_dev(a), to_dev(b), to_dev(c)
        plan.mul_assign_normalize_batch(da, db)
        want = a.copy()
        ref.mul_assign_normalize(want, b)
        assert np.array_equal(to_host(da, plan.dtype), want)
        da = to_dev(a)
        plan.normalize_batch(da)
        want = a.copy()
        ref.normalize(want)
        assert np.array_equal(to_host(da, plan.dtype), want)
        plan.mul_accumulate_batch(dc, to_dev(a), db)
        want = c.copy()
        ref.mul_accumulate(want, a, b)
        assert np.array_equal(to_host(dc, plan.dtype), want)


def test_test_product_property(oracle, plans):
    """The reference's own test_product (src/prime64.rs:1211-1267, src/prime32.rs:1006-1060):
    inv(fwd(a) (.) fwd(b)) == N * (a (*) b) and inv(mul_assign_normalize(fwd a, fwd b)) == a (*) b."""
    for bits, primes, sizes in ((64, U64_PRIMES, [16, 32, 64, 128, 256, 512, 1024]),
                                (32, U32_PRIMES, [32, 64, 128, 256, 512, 1024])):
        for p in primes:
            for n in sizes:
                plan = plans(bits, n, p)
                lhs = oracle.fill_uniform(n, p, 5 * n + 1, bits)
                rhs = oracle.fill_uniform(n, p, 5 * n + 2, bits)
                conv = oracle.negacyclic_convolution(n, p, lhs, rhs, bits)
                fl, fr = lhs.copy(), rhs.copy()
                plan.fwd(fl)
                plan.fwd(fr)
                assert int(fl.max()) < p and int(fr.max()) < p
                prod = fl.copy()
                plan.mul_assign_normalize(prod, fr)
                plan.inv(prod)
                assert int(prod.max()) < p
                assert np.array_equal(prod, conv), (bits, p, n)


@pytest.mark.parametrize("bits,n,p", [(64, 32, P62), (64, 64, P62), (64, 256, P62), (64, 1024, P62), (64, 2048, P62),
                                      (64, 1024, 9223372036853661697), (64, 512, 18446744069414584321),
                                      (64, 1024, 18446744073707716609), (64, 16, P62),
                                      (32, 64, 1062862849), (32, 1024, 1062862849), (32, 1024, 2147352577),
                                      (32, 512, 4293918721), (32, 4096, 1062862849), (32, 32, 1073479681),
                                      # 32-bit words above n = 4096: one polynomial per workgroup (mul_kernel_one), every class
                                      (32, 8192, 1062862849), (32, 8192, 2147352577), (32, 8192, 4293918721),
                                      (32, 16384, 1062862849), (32, 16384, 4293918721), (32, 16384, 2147352577),
                                      (32, 32768, 1062862849), (32, 32768, 2147352577), (32, 32768, 4293918721),
                                      (32, 65536, 1062862849)])
def test_fused_mul_ntt_equals_three_calls(oracle, plans, oplans, bits, n, p):
    """cntt_prime*_mul_ntt_batch == fwd; mul_assign_normalize; inv (src/prime64.rs:1254-1266), ragged batches,
    the fused kernels and the three-launch fallback (n = 65536)."""
    plan, ref = plans(bits, n, p), oplans(bits, n, p)
    for batch in (1, 13, 301):
        a = oracle.fill_uniform(batch * n, p, 31 + batch, bits)
        b = oracle.fill_uniform(batch * n, p, 97 + batch, bits)
        want, bn = a.copy(), b.copy()
        ref.fwd_batch(bn, 4)
        ref.fwd_batch(want, 4)
        ref.mul_assign_normalize(want, bn)
        ref.inv_batch(want, 4)
        da = to_dev(a)
        plan.mul_ntt_batch(da, to_dev(bn))
        got = to_host(da, plan.dtype)
        bad = np.nonzero(got != want)[0]
        assert bad.size == 0, "bits=%d n=%d p=%d batch=%d: %d mismatches, first at %d" % (bits, n, p, batch, bad.size,
                                                                                         bad[0])
        # and it is the negacyclic product of a and b
        if n <= 256 and batch == 1:
            assert np.array_equal(got, oracle.negacyclic_convolution(n, p, a, b, bits))


@pytest.mark.parametrize("bits,n,p", [(32, 256, 2127586817), (64, 256, 8762203435012018177)])
def test_strict_class_keeps_the_reference_barrett_wrap(oracle, plans, oplans, bits, n, p):
    """Primes of the strict class above 2^B / 3: the reference's Barrett product wraps for about 7e-4 of uniform operands on these two
    (src/prime32.rs:398-401, src/prime64.rs:549-552; tests/test_oracle_properties.py::test_oracle_keeps_the_reference_barrett_wrap) and
    its result is then NOT a b / n mod p.  Every device path -- the pointwise kernels, the fused product, the fused mul_accumulate
    chain -- has to return the reference's words, not the exact ones (a Montgomery product in the fused kernels, tried in round 4,
    returned the exact ones: found by tools/soak_random.py)."""
    run_wrap_case(oracle, plans(bits, n, p), oplans(bits, n, p), bits, n, p, 64, True)


@pytest.mark.parametrize("bits,p", [(64, 8762203435012018177), (64, 9223372036853661697), (32, 2127586817), (32, 2147352577),
                                    (32, 1944588929)])
def test_strict_class_transforms_mirror_the_reference_on_any_word(oracle, bits, p):
    """The strict class (31- / 63-bit primes) runs the reference's butterflies operation for operation (src/prime64/less_than_63bit.rs:117-154,
    214-232, src/prime32/less_than_31bit.rs; csrc/ntt_arith.hpp Bfly<T, CLS_STRICT>), because the reference's own mul_accumulate can hand `inv`
    a word above p for these primes.  Consequence, checked here: fwd and inv return the ORACLE'S words for ANY input word -- below 2p
    or anywhere in the B-bit range, far outside the API's contract -- on every kernel family (single pass, persistent walk, wave blocks, one
    polynomial per workgroup).  (The lazy classes are specified on canonical inputs only.)"""
    from concrete_ntt_amd import prime32, prime64
    mod = prime64 if bits == 64 else prime32
    seen = 0
    for n in ((16, 32, 256, 1024, 4096, 16384) if bits == 64 else (32, 256, 1024, 4096, 16384, 32768)):
        plan, ref = mod.Plan.try_new(n, p), oracle.Plan.try_new(n, p, bits)
        assert (plan is None) == (ref is None)
        if plan is None:
            continue
        for bound in (2 * p, 0):      # 0: raw B-bit words
            a = oracle.fill_uniform(8 * n, bound if bound < (1 << bits) else 0, 77 + n, bits)
            for name in ("fwd", "inv"):
                want = a.copy()
                getattr(ref, name + "_batch")(want, 4)
                d = to_dev(a)
                getattr(plan, name + "_batch")(d)
                assert np.array_equal(to_host(d, plan.dtype), want), (name, bits, n, p, bound)
                seen += 1
    assert seen >= 4


LAZY_ABOVE_POW2 = [(32, 536903681), (32, 268460033), (64, 2305843009214414849), (64, 1152921504607338497)]


@pytest.mark.parametrize("bits,p", LAZY_ABOVE_POW2)
def test_lazy_class_transforms_take_words_below_two_p(oracle, bits, p):
    """Lazy class (p < 2^(B-2)), primes just ABOVE a power of two: the reference's mul_accumulate can leave a word in [p, 2p) there (its
    Barrett estimate reaches 2p), and its butterflies reduce in every stage, the first included (src/prime64/
    less_than_62bit.rs:117-154,271-310), so fwd / inv of such words still return the canonical residues.  Every kernel family (single pass,
    persistent walk, wave blocks, one polynomial per workgroup) must do the same: words uniform in [0, 2p) in, the oracle's words out."""
    from concrete_ntt_amd import prime32, prime64
    mod = prime64 if bits == 64 else prime32
    seen = 0
    for n in ((16, 64, 1024, 2048, 4096, 16384) if bits == 64 else (32, 256, 1024, 2048, 4096)):
        plan, ref = mod.Plan.try_new(n, p), oracle.Plan.try_new(n, p, bits)
        assert (plan is None) == (ref is None)
        if plan is None:
            continue
        a = oracle.fill_uniform(8 * n, 2 * p, 177 + n, bits)
        assert int(a.max()) >= p
        for name in ("fwd", "inv"):
            want = a.copy()
            getattr(ref, name + "_batch")(want, 4)
            assert int(want.max()) < p
            d = to_dev(a)
            getattr(plan, name + "_batch")(d)
            assert np.array_equal(to_host(d, plan.dtype), want), (name, bits, n, p)
            seen += 1
    assert seen >= 4


@pytest.mark.parametrize("bits,n,p", [(32, 1024, 536903681), (32, 4096, 536903681)])
def test_lazy_class_fwd_mul_accumulate_inv_with_noncanonical_accumulators(oracle, plans, oplans, bits, n, p):
    """fwd -> mul_accumulate -> inv through the separate calls and through the composed / fused external_product, on primes where the
    reference's mul_accumulate really does hand `inv` words in [p, 2p) (counted here on the oracle: the case is exercised, not assumed)."""
    plan, ref = plans(bits, n, p), oplans(bits, n, p)
    batch = 256
    a = oracle.fill_uniform(batch * n, p, 3, bits)
    key = oracle.fill_uniform(batch * n, p, 4, bits)       # NTT-domain words
    acc = oracle.fill_uniform(batch * n, p, 5, bits)
    fa = a.copy()
    ref.fwd_batch(fa, 4)
    want = acc.copy()
    ref.mul_accumulate(want, fa, key)
    assert int((want >= want.dtype.type(p)).sum()) > 0, "this prime / seed no longer produces a non-canonical accumulator"
    wacc = want.copy()
    ref.inv_batch(want, 4)
    da, dk, dacc = to_dev(a), to_dev(key), to_dev(acc)
    plan.fwd_batch(da)
    plan.mul_accumulate_batch(dacc, da, dk)
    assert np.array_equal(to_host(dacc, plan.dtype), wacc)
    plan.inv_batch(dacc)
    assert np.array_equal(to_host(dacc, plan.dtype), want)
    # the chain entry point: nout = 5 takes the composed path (separate accumulate kernel + stand-alone inverse), nout = 2 the fused one
    for J, O in ((2, 5), (2, 2), (3, 1)):
        nb = 24
        terms = a[: nb * J * n]
        k = key[: J * O * n]
        ft = terms.copy()
        ref.fwd_batch(ft, 4)
        wout = np.zeros(nb * O * n, dtype=a.dtype)
        for e in range(nb):
            for o in range(O):
                acc1 = np.zeros(n, dtype=a.dtype)
                for j in range(J):
                    ref.mul_accumulate(acc1, np.ascontiguousarray(ft[(e * J + j) * n:(e * J + j + 1) * n]),
                                       np.ascontiguousarray(k[(j * O + o) * n:(j * O + o + 1) * n]))
                ref.inv(acc1)
                wout[(e * O + o) * n:(e * O + o + 1) * n] = acc1
        dout = to_dev(np.zeros(nb * O * n, dtype=a.dtype))
        plan.external_product_batch(dout, to_dev(terms), to_dev(k), J, O, False)
        assert np.array_equal(to_host(dout, plan.dtype), wout), (J, O)


def run_wrap_case(oracle, plan, ref, bits, n, p, batch, require_wrap, seed=0):
    """(also driven with random strict-class primes above 2^B / 3 by tools/soak_random.py wrap)"""
    a = oracle.fill_uniform(batch * n, p, 3 + seed, bits)
    bn = oracle.fill_uniform(batch * n, p, 4 + seed, bits)      # taken as NTT-domain words
    # mul_assign_normalize: the wrap is there (the oracle differs from the exact product) and the device reproduces it
    want = a.copy()
    ref.mul_assign_normalize(want, bn)
    n_inv = pow(n, -1, p)
    if require_wrap:
        assert any(int(want[i]) != int(a[i]) * int(bn[i]) * n_inv % p for i in range(batch * n))
    d = to_dev(a)
    plan.mul_assign_normalize_batch(d, to_dev(bn))
    assert np.array_equal(to_host(d, plan.dtype), want)
    # mul_accumulate
    acc = oracle.fill_uniform(batch * n, p, 5, bits)
    wacc = acc.copy()
    ref.mul_accumulate(wacc, a, bn)
    dacc = to_dev(acc)
    plan.mul_accumulate_batch(dacc, to_dev(a), to_dev(bn))
    assert np.array_equal(to_host(dacc, plan.dtype), wacc)
    # fused product: lhs in the coefficient domain
    wf = a.copy()
    ref.fwd_batch(wf, 4)
    exact_inputs = wf.copy()
    ref.mul_assign_normalize(wf, bn)
    if require_wrap:
        assert any(int(wf[i]) != int(exact_inputs[i]) * int(bn[i]) * n_inv % p for i in range(batch * n))
    ref.inv_batch(wf, 4)
    d = to_dev(a)
    plan.mul_ntt_batch(d, to_dev(bn))
    assert np.array_equal(to_host(d, plan.dtype), wf)
    # fused chain: J = 2 terms per element, O = 2 outputs
    J, O, nb = 2, 2, batch // 2
    key = bn[: J * O * n]
    wout = np.zeros(nb * O * n, dtype=a.dtype)
    ta = a.copy()
    ref.fwd_batch(ta, 4)
    # The reference's mul_accumulate reduces its Barrett remainder estimate (in [0, 3p)) by ONE conditional subtraction: an estimate in
    # [2p, 2^B) -- rarer still than the wrap -- leaves an accumulator in [p, 2^B - p), and the reference's inverse transform, whose butterflies
    # assume canonical words (src/prime64/less_than_63bit.rs:214-232), then returns what its instruction sequence happens to give.  The device's
    # strict-class butterflies are that instruction sequence (round 4), so those polynomials are compared like all others; their number is
    # returned for the record.
    noncanonical = 0
    for e in range(nb):
        for o in range(O):
            acc1 = np.zeros(n, dtype=a.dtype)
            for j in range(J):
                ref.mul_accumulate(acc1, ta[(e * J + j) * n:(e * J + j + 1) * n], np.ascontiguousarray(key[(j * O + o) * n:(j * O + o + 1) * n]))
            noncanonical += bool((acc1 >= a.dtype.type(p)).any())
            ref.inv(acc1)
            wout[(e * O + o) * n:(e * O + o + 1) * n] = acc1
    dout = to_dev(np.zeros(nb * O * n, dtype=a.dtype))
    plan.external_product_batch(dout, to_dev(a), to_dev(key), J, O, False)
    got = to_host(dout, plan.dtype).reshape(nb * O, n)
    bad = np.nonzero((got != wout.reshape(nb * O, n)).any(axis=1))[0]
    assert bad.size == 0, ("chain outputs differing from the oracle", bad[:8], noncanonical)
    # the same through the separate calls (composed path: stand-alone inverse kernel on the accumulators)
    dacc2 = to_dev(np.zeros(nb * O * n, dtype=a.dtype))
    dta = to_dev(a)
    plan.fwd_batch(dta)
    import torch
    t3, k3, a3 = dta.view(nb, J, n), to_dev(key).view(J, O, n), dacc2.view(nb, O, n)
    for o in range(O):
        acc = torch.zeros(nb * n, dtype=dta.dtype, device="cuda")
        for j in range(J):
            plan.mul_accumulate_batch(acc, t3[:, j, :].contiguous().view(-1), k3[j, o].repeat(nb))
        plan.inv_batch(acc)
        a3[:, o, :] = acc.view(nb, n)
    assert np.array_equal(to_host(dacc2, plan.dtype), wout)
    return noncanonical


P50, P51, P63, SOLINAS, PM64 = 1125899904679937, 2251799813554177, 9223372036853661697, 18446744069414584321, 18446744073707716609


@pytest.mark.parametrize("n,p", [(4096, P62), (8192, P62), (16384, P62), (4096, P50), (8192, P50), (16384, P50), (4096, P51),
                                 (16384, P51), (4096, P63), (8192, P63), (16384, P63), (4096, SOLINAS), (8192, PM64),
                                 (16384, SOLINAS), (16384, PM64), (32768, P62), (32768, P50), (32768, P51), (32768, P63),
                                 (32768, SOLINAS), (32768, PM64)])
def test_fused_mul_ntt_large_sizes(oracle, plans, oplans, n, p):
    """The fused product on the wave-block walk (mul_kernel_blk, u64 n = 4096 ... 16384; mul_kernel_32k at n = 32768: one
    register stage around two half walks; every class but the Montgomery one, which keeps the three launches): same values
    as fwd; mul_assign_normalize; inv of the oracle -- with more polynomials than resident workgroups (several trips round
    the persistent loop, prefetch included) and ragged batch sizes."""
    plan, ref = plans(64, n, p), oplans(64, n, p)
    for batch in (1, 5, 1100 if n == 4096 else 530 if n == 8192 else 270 if n == 16384 else 260):
        a = oracle.fill_uniform(batch * n, p, 131 + batch, 64)
        b = oracle.fill_uniform(batch * n, p, 197 + batch, 64)
        want, bn = a.copy(), b.copy()
        ref.fwd_batch(bn, 8)
        ref.fwd_batch(want, 8)
        ref.mul_assign_normalize(want, bn)
        ref.inv_batch(want, 8)
        da = to_dev(a)
        plan.mul_ntt_batch(da, to_dev(bn))
        got = to_host(da, plan.dtype)
        bad = np.nonzero(got != want)[0]
        assert bad.size == 0, "n=%d p=%d batch=%d: %d mismatches, first at %d" % (n, p, batch, bad.size, bad[0])


@pytest.mark.parametrize("n,p", [(16384, 1062862849), (32768, 1062862849), (32768, 2147352577), (16384, 4293918721)])
def test_fused_mul_ntt_large_sizes_u32(oracle, plans, oplans, n, p):
    """32-bit words on the wave-block walk (round 4: mul_kernel_blk<u32>, 2048-word blocks; the 30-bit class at n = 16384 with the
    asynchronous prefetch of the next polynomial, the others with ordinary loads): more polynomials than two rounds of the resident grid
    (two 512-thread / one 1024-thread workgroup per CU) and a ragged tail, against fwd; mul_assign_normalize; inv of the oracle."""
    plan, ref = plans(32, n, p), oplans(32, n, p)
    for batch in (3, 1100 if n == 16384 else 560):
        a = oracle.fill_uniform(batch * n, p, 131 + batch, 32)
        b = oracle.fill_uniform(batch * n, p, 197 + batch, 32)
        want, bn = a.copy(), b.copy()
        ref.fwd_batch(bn, 8)
        ref.fwd_batch(want, 8)
        ref.mul_assign_normalize(want, bn)
        ref.inv_batch(want, 8)
        da = to_dev(a)
        plan.mul_ntt_batch(da, to_dev(bn))
        got = to_host(da, plan.dtype)
        bad = np.nonzero(got != want)[0]
        assert bad.size == 0, "n=%d p=%d batch=%d: %d mismatches, first at %d" % (n, p, batch, bad.size, bad[0])


NATIVE = {"native32_plan32": native32.Plan32, "native64_plan32": native64.Plan32,
          "native128_plan32": native128.Plan32, "native_binary32_plan32": native_binary32.Plan32,
          "native_binary64_plan32": native_binary64.Plan32, "native_binary128_plan32": native_binary128.Plan32,
          "native32_plan52": native32.Plan52, "native64_plan52": native64.Plan52,
          "native_binary32_plan52": native_binary32.Plan52, "native_binary64_plan52": native_binary64.Plan52}


def _native_inputs(oracle, kind, n, seed, binary):
    ref = oracle.Native(kind, n)
    lhs, rhs = ref.words(), ref.words()
    raw = oracle.fill_uniform(lhs.size, 0, seed, 64)
    raw2 = oracle.fill_uniform(rhs.size, 0, seed + 99991, 64)
    lhs[:] = raw.astype(lhs.dtype) if lhs.dtype == np.uint64 else (raw >> np.uint64(32)).astype(np.uint32)
    rhs[:] = raw2.astype(rhs.dtype) if rhs.dtype == np.uint64 else (raw2 >> np.uint64(32)).astype(np.uint32)
    if binary:
        if ref.word == 16:
            rhs[0::2] &= np.uint64(1)
            rhs[1::2] = 0
        else:
            rhs &= rhs.dtype.type(1)
    return ref, lhs, rhs


@pytest.mark.parametrize("kind", sorted(NATIVE))
def test_native_polymul_golden(golden, oracle, kind):
    L = oracle.lib()
    for ent in [e for e in golden["polymul"] if e["kind"] == kind]:
        n, seed, wb = ent["n"], ent["seed"], ent["wordbits"]
        plan = NATIVE[kind].try_new(n)
        if wb == 128:
            lhs = np.empty(2 * n, dtype=np.uint64)
            rhs = np.empty(2 * n, dtype=np.uint64)
            for i in range(n):
                lhs[2 * i + 1], lhs[2 * i] = L.orc_splitmix64(seed + 2 * i), L.orc_splitmix64(seed + 2 * i + 1)
                rhs[2 * i + 1], rhs[2 * i] = (L.orc_splitmix64(seed + 7777 + 2 * i),
                                              L.orc_splitmix64(seed + 7778 + 2 * i))
            if ent["binary"]:
                rhs[0::2] &= np.uint64(1)
                rhs[1::2] = 0
        else:
            dt = np.uint64 if wb == 64 else np.uint32
            lhs = np.array([L.orc_splitmix64(seed + i) & ((1 << wb) - 1) for i in range(n)], dtype=dt)
            rhs = np.array([L.orc_splitmix64(seed + 7777 + i) & ((1 << wb) - 1) for i in range(n)], dtype=dt)
            if ent["binary"]:
                rhs &= dt(1)
        prod = np.zeros_like(lhs)
        plan.negacyclic_polymul(prod, lhs, rhs)
        assert sha(prod) == ent["prod_sha256"], (kind, n)


@pytest.mark.parametrize("kind", sorted(NATIVE))
def test_native_vs_oracle(oracle, kind):
    """fwd / fwd_binary / inv residues, round trip (src/native64.rs:1176-1206) and batched polymul
    (src/native64.rs:1208-1243) against the oracle; N up to 4096 (config C3's size)."""
    cls = NATIVE[kind]
    for n in (32, 64, 256, 1024, 2048, 4096):
        plan = cls.try_new(n)
        ref, lhs, rhs = _native_inputs(oracle, kind, n, 4242 + n, cls.BINARY)
        res_g = [np.zeros(n, dtype=plan.res_dtype) for _ in range(cls.NPRIMES)]
        res_o = ref.residues()
        plan.fwd(lhs, *res_g)
        ref.fwd(lhs, res_o)
        for a, b in zip(res_g, res_o):
            assert np.array_equal(a, b), (kind, n, "fwd")
        if cls.BINARY:
            rb_g = [np.zeros(n, dtype=plan.res_dtype) for _ in range(cls.NPRIMES)]
            rb_o = ref.residues()
            plan.fwd_binary(rhs, *rb_g)
            ref.fwd_binary(rhs, rb_o)
            for a, b in zip(rb_g, rb_o):
                assert np.array_equal(a, b), (kind, n, "fwd_binary")
        out_g, out_o = np.zeros_like(lhs), np.zeros_like(lhs)
        plan.inv(out_g, *res_g)
        ref.inv(out_o, res_o)
        assert np.array_equal(out_g, out_o), (kind, n, "inv")
        for a, b in zip(res_g, res_o):  # inv also overwrites the residue buffers: src/native64.rs:1010-1014
            assert np.array_equal(a, b), (kind, n, "inv residues")
        # round trip == value * n (wrapping): src/native64.rs:1204-1206
        if ref.word != 16:
            assert np.array_equal(out_g, lhs * lhs.dtype.type(n))
    # batched polymul, device resident
    n, batch = 1024, 5
    plan = cls.try_new(n)
    ref = oracle.Native(kind, n)
    wpp = n * (2 if ref.word == 16 else 1)
    lhs = np.concatenate([_native_inputs(oracle, kind, n, 9000 + b, cls.BINARY)[1] for b in range(batch)])
    rhs = np.concatenate([_native_inputs(oracle, kind, n, 9000 + b, cls.BINARY)[2] for b in range(batch)])
    want = np.zeros_like(lhs)
    ref.negacyclic_polymul_batch(want, lhs, rhs, batch, 4)
    dl, dr = to_dev(lhs), to_dev(rhs)
    dp = to_dev(np.zeros_like(lhs))
    plan.negacyclic_polymul_batch(dp, dl, dr)
    got = to_host(dp, lhs.dtype)
    assert np.array_equal(got, want), kind
    assert wpp * batch == lhs.size


def test_native_crt_golden(golden, oracle):
    """CRT kernel on arbitrary residues: feed constant-term polynomials so that inv() lands each
    golden residue vector at index 0 (truth: mixed-radix digits with the top-digit sign rule)."""
    for grp in golden["crt"]:
        kind = grp["kind"]
        cls = NATIVE[kind]
        n = 32
        plan = cls.try_new(n)
        for v in grp["vectors"]:
            res = []
            for k, r in enumerate(v["residues"]):
                sub = plan.ntt(k)
                p, ninv = sub.modulus(), sub.info().n_inv_mod_p
                buf = np.zeros(n, dtype=plan.res_dtype)
                buf[0] = (r * ninv) % p  # fwd of a constant c is c everywhere; inv of that is n*c at index 0
                sub.fwd(buf)
                res.append(buf)
            out = np.zeros(n * (2 if cls.WORD == 16 else 1), dtype=plan.word_dtype)
            plan.inv(out, *res)
            got = int(out[0]) | (int(out[1]) << 64) if cls.WORD == 16 else int(out[0])
            assert got == v["value"], (kind, v)


def test_length_mismatch_is_a_panic(plans):
    plan = plans(64, 1024, P62)
    with pytest.raises(cntt.Panic):
        plan.fwd(np.zeros(512, dtype=np.uint64))
    nat = native64.Plan32.try_new(64)
    with pytest.raises(cntt.Panic):  # assert_eq!(n, lhs.len()): src/native64.rs:1043-1045
        nat.negacyclic_polymul(np.zeros(64, dtype=np.uint64), np.zeros(32, dtype=np.uint64),
                               np.zeros(64, dtype=np.uint64))


def test_empty_batch_is_a_noop(plans):
    torch = _torch()
    plan = plans(64, 1024, P62)
    plan.fwd_batch(torch.empty(0, dtype=torch.int64, device="cuda"))
    plan.inv_batch(np.zeros(0, dtype=np.uint64))


# ---- BASELINE.json full sizes: size-independent properties + sampled oracle comparison -----------
def test_c2_full_size_properties(oracle, plans, oplans):
    """configs[1]: prime64 N=1024 batch=65536, fwd + pointwise + inv on device-resident data."""
    torch = _torch()
    n, batch = 1024, 65536
    plan, ref = plans(64, n, P62), oplans(64, n, P62)
    a = torch.empty(batch * n, dtype=torch.int64, device="cuda")
    b = torch.empty(batch * n, dtype=torch.int64, device="cuda")
    cntt.fill_uniform(a, P62, 0x5EED0002, )
    cntt.fill_uniform(b, P62, 0x5EED1002)
    a0 = a.clone()
    # sampled polynomials against the oracle (inputs regenerated on the CPU from the same seeds)
    sample = [0, 1, 4097, 32768, 65535]
    plan.fwd_batch(a)
    fa = a.clone()
    for s in sample:
        want = oracle.fill_uniform(n, P62, 0x5EED0002 + s * n, 64)
        assert np.array_equal(to_host(a0[s * n:(s + 1) * n], np.uint64), want)
        ref.fwd(want)
        assert np.array_equal(to_host(a[s * n:(s + 1) * n], np.uint64), want), s
    # round trip: normalize(inv(fwd(x))) == x for the whole batch
    plan.inv_batch(a)
    plan.normalize_batch(a)
    assert torch.equal(a, a0)
    # polymul through the NTT domain equals the oracle's on the sampled polynomials
    plan.fwd_batch(b)
    plan.mul_assign_normalize_batch(fa, b)
    plan.inv_batch(fa)
    for s in sample:
        x = oracle.fill_uniform(n, P62, 0x5EED0002 + s * n, 64)
        y = oracle.fill_uniform(n, P62, 0x5EED1002 + s * n, 64)
        ref.fwd(x)
        ref.fwd(y)
        ref.mul_assign_normalize(x, y)
        ref.inv(x)
        assert np.array_equal(to_host(fa[s * n:(s + 1) * n], np.uint64), x), s
    # canonical outputs
    assert int(fa.max()) < P62 and int(fa.min()) >= 0
    # the fused kernel (cntt_prime64_mul_ntt_batch) gives the same 64 Mi words as the three launches
    c = a0.clone()
    plan.mul_ntt_batch(c, b)
    assert torch.equal(c, fa)


def test_c4_shard_size_properties(oracle, plans, oplans):
    """configs[3] per-GPU slice: prime64 N=16384 (one polynomial per workgroup, 128 KiB of LDS),
    a 2048-polynomial piece of the 131072-polynomial shard; round trip + sampled oracle comparison."""
    torch = _torch()
    n, batch = 16384, 2048
    plan, ref = plans(64, n, P62), oplans(64, n, P62)
    a = torch.empty(batch * n, dtype=torch.int64, device="cuda")
    cntt.fill_uniform(a, P62, 0x5EED0004)
    a0 = a.clone()
    plan.fwd_batch(a)
    for s in (0, 1023, 2047):
        want = oracle.fill_uniform(n, P62, 0x5EED0004 + s * n, 64)
        ref.fwd(want)
        assert np.array_equal(to_host(a[s * n:(s + 1) * n], np.uint64), want), s
    plan.inv_batch(a)
    plan.normalize_batch(a)
    assert torch.equal(a, a0)


def test_c3_c5_native_full_n(oracle):
    """configs[2] native64 N=4096 and configs[4] native_binary64 N=2048: batched device polymul on a
    256-polynomial slice, every polynomial against the oracle (whose truth is pinned by the schoolbook
    convolution in tests/test_oracle_golden.py)."""
    for kind, n in (("native64_plan32", 4096), ("native_binary64_plan32", 2048)):
        cls = NATIVE[kind]
        batch = 256
        plan = cls.try_new(n)
        ref = oracle.Native(kind, n)
        lhs = oracle.fill_uniform(batch * n, 0, 0x5EED0003, 64)
        rhs = oracle.fill_uniform(batch * n, 0, 0x5EED1003, 64)
        if cls.BINARY:
            rhs &= np.uint64(1)
        want = np.zeros_like(lhs)
        ref.negacyclic_polymul_batch(want, lhs, rhs, batch, 8)
        dp = to_dev(np.zeros_like(lhs))
        plan.negacyclic_polymul_batch(dp, to_dev(lhs), to_dev(rhs))
        assert np.array_equal(to_host(dp, np.uint64), want), kind


@pytest.mark.parametrize("kind", ["native32_plan52", "native64_plan52", "native_binary32_plan52", "native_binary64_plan52"])
def test_plan52_polymul_through_the_plan32_kernel_equals_the_50_bit_pipeline(oracle, kind):
    """Round 5: negacyclic_polymul of a Plan52 kind runs the Plan32 whole-product kernel of the same words (the wrapping product does not
    depend on the primes: src/native64.rs:1074-1165 against :1042-1069 return the same words).  Here: that path == the composed pipeline on
    the 50-bit primes (testing switch plan52_via32 = 0) == the oracle's Plan52, for sizes on both sides of every kernel shape, extremes
    included; n = 16 (no Plan32 plan exists there: prime32 needs n >= 32) takes the composed path by itself."""
    torch = _torch()
    cls = NATIVE[kind]
    for n in (16, 32, 1024, 4096, 8192, 32768):
        plan = cls.try_new(n)
        ref, lhs, rhs = _native_inputs(oracle, kind, n, 777 + n, cls.BINARY)
        batch = 5
        lhs = np.concatenate([lhs] * batch)
        rhs = np.concatenate([rhs] * batch)
        lhs[:n] = np.iinfo(lhs.dtype).max                      # one all-ones product: the bound of |c|
        if not cls.BINARY:
            rhs[:n] = np.iinfo(rhs.dtype).max
        want = np.zeros_like(lhs)
        ref.negacyclic_polymul_batch(want, lhs, rhs, batch, 4)
        sdt = np.int64 if lhs.dtype == np.uint64 else np.int32
        outs = []
        for via in (1, 0):
            with cntt.debug_switches(plan52_via32=via):
                dp = torch.zeros(lhs.size, dtype=torch.int64 if sdt == np.int64 else torch.int32, device="cuda")
                plan.negacyclic_polymul_batch(dp, torch.from_numpy(lhs.view(sdt)).cuda(), torch.from_numpy(rhs.view(sdt)).cuda())
                outs.append(dp.cpu().numpy().view(lhs.dtype))
        assert np.array_equal(outs[0], want), (kind, n, "through the Plan32 kernel")
        assert np.array_equal(outs[1], want), (kind, n, "composed on the 50-bit primes")


def _composed_polymul(torch, plan, cls, dl, dr, batch, n):
    """split -> per-prime transforms -> pointwise -> inverse -> CRT through the plan's separate entry points"""
    rl = [torch.empty(batch * n, dtype=torch.int32, device="cuda") for _ in range(cls.NPRIMES)]
    rr = [torch.empty(batch * n, dtype=torch.int32, device="cuda") for _ in range(cls.NPRIMES)]
    plan.fwd_batch(dl, rl)
    plan.fwd_batch(dr, rr, binary=cls.BINARY)
    for i in range(cls.NPRIMES):
        plan.ntt(i).mul_assign_normalize_batch(rl[i], rr[i])
    composed = torch.empty_like(dl)
    plan.inv_batch(composed, rl)
    return composed


@pytest.mark.parametrize("n", [32, 256, 2048, 4096])
def test_native128_polymul_persistent_kernel(oracle, n):
    """native128::Plan32 (ten primes) runs the persistent whole-product kernel at every size, 4096 / n products per
    workgroup.  A batch of more than 2 x 256 x 4096 / n products, not a multiple of the group: every workgroup loops and the
    last group is ragged.  Samples against the oracle, every product against the composed pipeline."""
    torch = _torch()
    kind = "native128_plan32"
    cls = NATIVE[kind]
    plan, ref = cls.try_new(n), oracle.Native(kind, n)
    batch = 2 * 256 * (4096 // n) + 4096 // n + 3
    wpp = 2 * n
    dl = torch.empty(batch * wpp, dtype=torch.int64, device="cuda")
    dr = torch.empty_like(dl)
    cntt.fill_uniform(dl, 0, 0x128A)
    cntt.fill_uniform(dr, 0, 0x128B)
    dp = torch.empty_like(dl)
    plan.negacyclic_polymul_batch(dp, dl, dr)
    for b in (0, 1, 4096 // n - 1, 4096 // n, batch // 2, batch - 2, batch - 1):
        lhs = to_host(dl[b * wpp:(b + 1) * wpp], np.uint64).copy()
        rhs = to_host(dr[b * wpp:(b + 1) * wpp], np.uint64).copy()
        want = np.zeros(wpp, dtype=np.uint64)
        ref.negacyclic_polymul(want, lhs, rhs)
        assert np.array_equal(to_host(dp[b * wpp:(b + 1) * wpp], np.uint64), want), (n, b)
    assert torch.equal(dp, _composed_polymul(torch, plan, cls, dl, dr, batch, n)), n


@pytest.mark.parametrize("n", [8192, 16384, 32768])
@pytest.mark.parametrize("kind", ["native32_plan32", "native64_plan32", "native128_plan32", "native_binary32_plan32",
                                  "native_binary64_plan32", "native_binary128_plan32"])
def test_native_polymul_large_n(oracle, kind, n):
    """N = 8192 (some kinds) / 16384 / 32768 run the persistent whole-product kernel whose workgroups park residue tiles
    in the plan's workspace (csrc/native_fused.hpp, native_polymul_kernel_g; 32 coefficients per thread at 32768).  A batch larger than the grid (every workgroup loops, the
    last round is ragged): the first, the last and a few middle products against the oracle, and every product against the
    composed split -> transforms -> pointwise -> CRT pipeline."""
    torch = _torch()
    cls = NATIVE[kind]
    plan, ref = cls.try_new(n), oracle.Native(kind, n)
    batch = 2 * 256 + 37
    wpp = n * (2 if ref.word == 16 else 1)
    per = [_native_inputs(oracle, kind, n, 31000 + b, cls.BINARY) for b in range(7)]
    lhs = np.concatenate([per[b % 7][1] if b % 5 else np.roll(per[b % 7][1], b) for b in range(batch)])
    rhs = np.concatenate([per[(b + 3) % 7][2] for b in range(batch)])
    dl, dr = to_dev(lhs), to_dev(rhs)
    dp = to_dev(np.zeros_like(lhs))
    plan.negacyclic_polymul_batch(dp, dl, dr)
    got = to_host(dp, lhs.dtype)
    for b in (0, 1, 255, 256, 511, 512, batch - 1):
        want = np.zeros(wpp, dtype=lhs.dtype)
        ref.negacyclic_polymul(want, lhs[b * wpp:(b + 1) * wpp].copy(), rhs[b * wpp:(b + 1) * wpp].copy())
        assert np.array_equal(got[b * wpp:(b + 1) * wpp], want), (kind, n, b)
    assert torch.equal(dp, _composed_polymul(torch, plan, cls, dl, dr, batch, n)), (kind, n)


@pytest.mark.parametrize("kind,n,batch", [("native64_plan32", 8192, 300), ("native128_plan32", 1024, 2100),
                                          ("native64_plan52", 1024, 64)])
def test_native_polymul_streams_and_threads_share_a_plan(kind, n, batch):
    """One plan, one workspace (parking area of the persistent kernel / residue arrays of the composed pipeline), used from
    two HIP streams and from two host threads at once: the library orders the calls that share the workspace
    (csrc/host.hip, Workspace), so every result equals the one computed alone."""
    import threading
    torch = _torch()
    cls = NATIVE[kind]
    plan = cls.try_new(n)
    wpp = n * (2 if cls.WORD == 16 else 1)
    dt = torch.int32 if cls.WORD == 4 else torch.int64
    sets = []
    for k in range(4):
        lhs = torch.empty(batch * wpp, dtype=dt, device="cuda")
        rhs = torch.empty_like(lhs)
        cntt.fill_uniform(lhs, 0, 0x57 + 2 * k)
        cntt.fill_uniform(rhs, 0, 0x58 + 2 * k)
        want = torch.empty_like(lhs)
        plan.negacyclic_polymul_batch(want, lhs, rhs)
        sets.append((lhs, rhs, want))
    torch.cuda.synchronize()
    # (1) two streams, calls interleaved from one thread
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    outs = [torch.zeros_like(sets[0][0]) for _ in range(4)]
    for rep in range(6):
        for k in range(4):
            with torch.cuda.stream(streams[k % 2]):
                plan.negacyclic_polymul_batch(outs[k], sets[k][0], sets[k][1])
    torch.cuda.synchronize()
    for k in range(4):
        assert torch.equal(outs[k], sets[k][2]), (kind, "streams", k)
    # (2) two host threads, each on its own stream
    outs = [torch.zeros_like(sets[0][0]) for _ in range(4)]
    errors = []

    def worker(t):
        try:
            with torch.cuda.stream(streams[t]):
                for rep in range(6):
                    for k in (t, t + 2):
                        plan.negacyclic_polymul_batch(outs[k], sets[k][0], sets[k][1])
        except Exception as e:  # noqa: BLE001
            errors.append(e)

    th = [threading.Thread(target=worker, args=(t,)) for t in range(2)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    torch.cuda.synchronize()
    assert not errors, errors
    for k in range(4):
        assert torch.equal(outs[k], sets[k][2]), (kind, "threads", k)


def test_batch_calls_capture_into_a_hip_graph(oracle):
    """Device-resident _batch calls only enqueue (no allocation, no synchronisation), so a launch-bound sequence can be
    captured once into a hipGraph and replayed: fwd -> mul_assign_normalize -> inv, the fused mul_ntt, and a native
    polymul whose workspace was reserved beforehand.  Replays are checked against the oracle."""
    torch = _torch()
    n, batch = 256, 6
    plan, oplan = prime64.Plan.try_new(n, P62), oracle.Plan.try_new(n, P62, 64)
    a = oracle.fill_uniform(batch * n, P62, 77, 64)
    b = oracle.fill_uniform(batch * n, P62, 78, 64)
    fb = b.copy()
    want = a.copy()
    for i in range(batch):
        oplan.fwd(fb[i * n:(i + 1) * n])
        oplan.fwd(want[i * n:(i + 1) * n])
    oplan.mul_assign_normalize(want, fb)
    for i in range(batch):
        oplan.inv(want[i * n:(i + 1) * n])
    nplan = native64.Plan32.try_new(n)
    nplan.reserve(batch)
    lhs = oracle.fill_uniform(batch * n, 0, 79, 64)
    rhs = oracle.fill_uniform(batch * n, 0, 80, 64)
    src, dfb = to_dev(a), to_dev(fb)
    x, y = torch.empty_like(src), torch.empty_like(src)
    dl, dr, dp = to_dev(lhs), to_dev(rhs), torch.empty_like(src)
    # warm every kernel and table upload outside the capture
    x.copy_(src); plan.fwd_batch(x); plan.mul_assign_normalize_batch(x, dfb); plan.inv_batch(x)
    y.copy_(src); plan.mul_ntt_batch(y, dfb)
    nplan.negacyclic_polymul_batch(dp, dl, dr)
    ref_native = to_host(dp, np.uint64).copy()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        x.copy_(src)
        plan.fwd_batch(x)
        plan.mul_assign_normalize_batch(x, dfb)
        plan.inv_batch(x)
        y.copy_(src)
        plan.mul_ntt_batch(y, dfb)
        nplan.negacyclic_polymul_batch(dp, dl, dr)
    for _ in range(3):
        x.zero_(); y.zero_(); dp.zero_()
        g.replay()
        torch.cuda.synchronize()
        assert np.array_equal(to_host(x, np.uint64), want)
        assert np.array_equal(to_host(y, np.uint64), want)
        assert np.array_equal(to_host(dp, np.uint64), ref_native)
    # the native product itself against the schoolbook wrapping convolution (polynomial 0)
    onat = oracle.Native("native64_plan32", n)
    prod0 = np.zeros(n, dtype=np.uint64)
    onat.negacyclic_polymul(prod0, lhs[:n].copy(), rhs[:n].copy())
    assert np.array_equal(ref_native[:n], prod0)


@pytest.mark.parametrize("kind,n", [("native64_plan32", 8192), ("native128_plan32", 1024), ("native64_plan52", 512)])
def test_native_polymul_with_workspace_captures_into_a_hip_graph(oracle, kind, n):
    """The native plans that need a workspace (parking area of the persistent kernel, residue arrays of the composed
    pipeline): after cntt_native_reserve the call neither allocates nor synchronises, so it captures; replays on fresh
    inputs are checked against the oracle."""
    torch = _torch()
    cls = NATIVE[kind]
    plan, ref = cls.try_new(n), oracle.Native(kind, n)
    batch = 5
    wpp = n * (2 if ref.word == 16 else 1)
    ins = [_native_inputs(oracle, kind, n, 41000 + b, cls.BINARY) for b in range(2 * batch)]
    plan.reserve(batch)
    dl = to_dev(np.concatenate([ins[b][1] for b in range(batch)]))
    dr = to_dev(np.concatenate([ins[b][2] for b in range(batch)]))
    dp = torch.zeros_like(dl)
    plan.negacyclic_polymul_batch(dp, dl, dr)   # warm-up: table uploads happen outside the capture
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        plan.negacyclic_polymul_batch(dp, dl, dr)
    for rnd in range(2):
        lhs = np.concatenate([ins[rnd * batch + b][1] for b in range(batch)])
        rhs = np.concatenate([ins[rnd * batch + b][2] for b in range(batch)])
        dl.copy_(to_dev(lhs))
        dr.copy_(to_dev(rhs))
        dp.zero_()
        g.replay()
        torch.cuda.synchronize()
        want = np.zeros_like(lhs)
        ref.negacyclic_polymul_batch(want, lhs, rhs, batch, 2)
        assert np.array_equal(to_host(dp, lhs.dtype), want), (kind, n, rnd)
    assert wpp * batch == dl.numel()


@pytest.mark.parametrize("bits,logn,p", [
    (64, 17, 4611686018425815041),      # lazy class, depth-3 global stages (62-bit prime = 1 mod 2^18)
    (64, 18, 18446744069414584321),     # Solinas (generic class), depth 4
    (64, 17, 9223372036836950017),      # strict class: largest 63-bit prime = 1 mod 2^18
    (32, 17, 2013265921),               # 15 * 2^27 + 1 (31-bit class), depth 2
    (32, 18, 4293918721),               # generic u32 (2^32 - 2^20 + 1), depth 3
])
def test_very_large_transforms_vs_oracle(oracle, plans, oplans, bits, logn, p):
    """N = 2^17, 2^18: several global stages in front of / behind the LDS-resident sub-transforms
    (the reference's depth-first recursion, src/prime64/shoup.rs:660-682), every arithmetic class."""
    n = 1 << logn
    if oracle.Plan.try_new(n, p, bits) is None:
        pytest.skip("%d is not an NTT prime for n = 2^%d" % (p, logn))
    _batch_case(oracle, plans, oplans, bits, n, p, 3, 7000 + logn)
