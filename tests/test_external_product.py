"""Fused mul_accumulate chain (cntt_prime*_external_product_batch) against the oracle's
fwd / mul_accumulate / inv called in sequence (src/prime64.rs:794, :1085-1128, :872).  Covers the fused kernels (persistent
walk with an LDS twiddle image up to n = 2048 u64 / 4096 u32, wave-block walk for u64 n = 4096 ... 16384, one element per
workgroup for u32 n = 8192 ... 32768; nout <= 4 -- three / four outputs of the largest size of each width as two launches
of <= 2 outputs), the composed path (single-pass sizes, nout = 5, the Montgomery class at large n), every arithmetic class,
ragged batches, accumulate mode and the empty sum.  Bit-exact."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

P62, P63, SOLINAS = 4611686018427322369, 9223372036853661697, 18446744069414584321
P30, P31, P32 = 1062862849, 2147352577, 4293918721


def _dev(a):
    import torch
    return torch.from_numpy(a.view(np.int64 if a.dtype == np.uint64 else np.int32).copy()).cuda()


def _host(t, dtype):
    return t.cpu().numpy().view(dtype)


def _expected(oplan, terms, key, init, n, J, O, batch, p, accumulate):
    dt = terms.dtype
    kf = key.reshape(J, O, n)
    out = np.zeros((batch, O, n), dtype=dt)
    for b in range(batch):
        acc = [np.zeros(n, dtype=dt) for _ in range(O)]
        for j in range(J):
            t = terms.reshape(batch, J, n)[b, j].copy()
            oplan.fwd(t)
            for o in range(O):
                oplan.mul_accumulate(acc[o], t, np.ascontiguousarray(kf[j, o]))
        for o in range(O):
            oplan.inv(acc[o])
            if accumulate:
                s = acc[o].astype(object) + init.reshape(batch, O, n)[b, o].astype(object)
                acc[o] = np.array([int(x) % p for x in s], dtype=dt)
            out[b, o] = acc[o]
    return out.reshape(-1)


CASES = [
    # bits, n, p, J, O, batch
    (64, 1024, P62, 6, 2, 5),        # fused, lazy class, the NTT-PBS-like shape (k = 1, three levels)
    (64, 1024, P62, 3, 4, 3),        # fused, four outputs
    (64, 256, P63, 2, 3, 9),         # fused, strict class, several polynomials per workgroup with a ragged tail
    (64, 512, SOLINAS, 4, 1, 3),     # fused, generic (Montgomery) class
    (64, 16, P62, 2, 2, 3),          # single-pass transform: composed path
    (64, 2048, P62, 3, 2, 2),        # two wavefronts per element
    (64, 1024, P62, 2, 5, 2),        # nout = 5: composed path
    (32, 1024, P30, 6, 2, 5),        # fused u32
    (32, 4096, P31, 2, 2, 3),        # fused u32, 32 registers per thread
    (32, 64, P32, 3, 3, 7),          # fused u32, generic class
    (32, 8192, P30, 2, 1, 2),        # u32 above n = 4096: one batch element per workgroup (ExtOne) ...
    (32, 8192, P31, 3, 4, 2),
    (32, 16384, P30, 3, 2, 3),
    (32, 16384, P32, 2, 3, 2),       # p >= 2^31 on doubles
    (32, 32768, P30, 2, 2, 2),
    (32, 32768, P32, 2, 1, 1),       # p >= 2^31 at this size (Montgomery class until round 3, doubles since)
    (32, 32768, P30, 2, 3, 1),       # ... three / four outputs at n = 32768: two fused launches of <= 2 outputs (round 4)
    (32, 32768, P30, 3, 4, 2),
    (32, 32768, P32, 2, 3, 1),       # p >= 2^31 at this size: on doubles since round 4
    # the chain on the wave-block walk (ExtBlk): u64 n = 4096 / 8192 (1 .. 4 outputs) and 16384 (1 .. 2), every class but the Montgomery one
    (64, 4096, P62, 3, 2, 3),
    (64, 4096, P62, 2, 1, 2),
    (64, 4096, P62, 2, 3, 2),
    (64, 4096, P63, 2, 4, 2),
    (64, 4096, 1125899904679937, 9, 2, 2),      # CLS_FP: more terms than the accumulator's reduction period
    (64, 4096, 2251799813554177, 3, 2, 2),      # CLS_FP51
    (64, 4096, SOLINAS, 2, 2, 2),               # 2^64 - c
    (64, 8192, P62, 2, 2, 2),
    (64, 8192, 1125899904679937, 2, 4, 1),
    (64, 8192, 18446744073707716609, 2, 1, 2),  # 2^64 - c
    (64, 16384, P62, 2, 1, 1),                  # n = 16384: the wave-block chain for one / two outputs (128 VGPRs) ...
    (64, 16384, P62, 3, 2, 2),
    (64, 16384, 1125899904679937, 9, 2, 2),     # CLS_FP: more terms than the accumulator's reduction period
    (64, 16384, SOLINAS, 2, 2, 1),
    (64, 16384, P63, 2, 1, 2),
    (64, 16384, P62, 2, 3, 1),                  # ... three / four outputs: two fused launches of <= 2 outputs (round 4)
    (64, 16384, P62, 3, 4, 2),
    (64, 16384, 1125899904679937, 9, 4, 1),     # CLS_FP, split launches, more terms than the accumulator's reduction period
    (64, 16384, SOLINAS, 2, 3, 2),
    (64, 16384, 9224497936763846657, 2, 3, 1),  # Montgomery class at this size: no fused kernel at all, composed path
    # three / four 64-bit accumulator tiles: the shapes that run without the next-term prefetch (ExtWp NEXT = false) ...
    (64, 2048, P62, 3, 3, 3),
    (64, 2048, P63, 2, 4, 2),
    (64, 512, SOLINAS, 3, 3, 5),                # 2^64 - c, three outputs
    (64, 1024, 18446744073707716609, 2, 4, 3),  # 2^64 - c, four outputs
    (64, 1024, 1125899904679937, 13, 4, 2),     # CLS_FP, four outputs, more terms than the accumulator's reduction period
    (64, 2048, 2251799813554177, 3, 4, 2),      # CLS_FP51
    (64, 1024, 9224497936763846657, 3, 4, 2),   # Montgomery class (p >= 2^63, not 2^64 - c)
    (64, 256, 9224497936763846657, 2, 3, 5),
    # ... and 32-bit words on the 16-coefficient schedules (four outputs; p >= 2^31 from two / three outputs on)
    (32, 2048, P30, 3, 4, 3),
    (32, 4096, P31, 2, 4, 2),
    (32, 2048, P32, 2, 2, 3),
    (32, 2048, P32, 5, 3, 2),
    (32, 4096, P32, 2, 3, 2),
    (32, 4096, P32, 2, 4, 1),
]


@pytest.mark.parametrize("bits,n,p,J,O,batch", CASES)
@pytest.mark.parametrize("accumulate", [False, True])
def test_gpu_external_product(oracle, bits, n, p, J, O, batch, accumulate):
    from concrete_ntt_amd import prime32, prime64
    mod = prime64 if bits == 64 else prime32
    plan, oplan = mod.Plan.try_new(n, p), oracle.Plan.try_new(n, p, bits)
    assert plan is not None and oplan is not None
    terms = oracle.fill_uniform(batch * J * n, p, 11 + n, bits)
    key = oracle.fill_uniform(J * O * n, p, 22 + n, bits)
    init = oracle.fill_uniform(batch * O * n, p, 33 + n, bits)
    want = _expected(oplan, terms, key, init, n, J, O, batch, p, accumulate)
    dt = np.uint64 if bits == 64 else np.uint32
    dout = _dev(init if accumulate else np.zeros(batch * O * n, dtype=dt))
    dterms, dkey = _dev(terms), _dev(key)
    plan.external_product_batch(dout, dterms, dkey, J, O, accumulate)
    got = _host(dout, dt)
    assert np.array_equal(got, want)
    assert np.array_equal(_host(dterms, dt), terms) and np.array_equal(_host(dkey, dt), key)  # inputs untouched
    # host-memory call: same result
    hout = (init if accumulate else np.zeros(batch * O * n, dtype=dt)).copy()
    plan.external_product_batch(hout, terms, key, J, O, accumulate)
    assert np.array_equal(hout, want)


def test_gpu_external_product_empty_sum_and_shapes(oracle):
    from concrete_ntt_amd import Panic, prime64
    n = 256
    plan = prime64.Plan.try_new(n, P62)
    out = oracle.fill_uniform(2 * 3 * n, P62, 5, 64)
    keep = out.copy()
    empty = np.zeros(0, dtype=np.uint64)
    plan.external_product_batch(out, empty, empty, 0, 3, True)   # nterms = 0, accumulate: unchanged
    assert np.array_equal(out, keep)
    plan.external_product_batch(out, empty, empty, 0, 3, False)  # nterms = 0, replace: zero
    assert not out.any()
    with pytest.raises(Panic):
        plan.external_product_batch(out, np.zeros(n, dtype=np.uint64), np.zeros(3 * n, dtype=np.uint64), 1, 3)


def test_gpu_external_product_is_sum_of_products(oracle):
    """Property at full size: with the key pre-normalised (key = fwd(g) * n^-1), the chain equals the sum of the
    negacyclic products t_j * g_j -- checked against the three-call mul_ntt path on a large batch."""
    import torch
    import concrete_ntt_amd as cntt
    from concrete_ntt_amd import prime64
    n, J, O, batch = 1024, 4, 2, 512
    plan = prime64.Plan.try_new(n, P62)
    terms = torch.empty(batch * J * n, dtype=torch.int64, device="cuda")
    cntt.fill_uniform(terms, P62, 1)
    g = torch.empty(J * O * n, dtype=torch.int64, device="cuda")
    cntt.fill_uniform(g, P62, 2)
    key = g.clone()
    plan.fwd_batch(key)
    plan.normalize_batch(key)
    out = torch.zeros(batch * O * n, dtype=torch.int64, device="cuda")
    plan.external_product_batch(out, terms, key, J, O)
    # reference composition on device: per (j, o) the fused single product, summed mod p with mul_accumulate-free adds
    kfull = g.clone()
    plan.fwd_batch(kfull)
    acc = np.zeros((batch, O, n), dtype=object)
    t3 = terms.view(batch, J, n)
    for j in range(J):
        for o in range(O):
            a = t3[:, j, :].contiguous().view(-1).clone()
            rhs = kfull.view(J, O, n)[j, o].repeat(batch)
            plan.mul_ntt_batch(a, rhs)
            acc[:, o, :] += a.cpu().numpy().view(np.uint64).reshape(batch, n).astype(object)
    want = np.array([int(x) % P62 for x in acc.reshape(-1)], dtype=np.uint64)
    assert np.array_equal(out.cpu().numpy().view(np.uint64), want)
