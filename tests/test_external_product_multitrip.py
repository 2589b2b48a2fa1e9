"""The PERSISTENT loops of the fused mul_accumulate chain kernels over more than two rounds of the resident grid
(VERDICT round 3): ext_kernel_wp and ext_kernel_blk launch min(CUs x PER_CU, tiles) workgroups and walk
`tile += gridDim.x`; every other chain test has a batch of at most 9, i.e. one trip.  The hand-over between one
element's last inverse read of the exchange buffer and the next element's first forward write is exercised only by
a second and third trip.

Every case: batch = 2 rounds of the grid + a ragged tail, compared
  * on EVERY element with the same step as separate batched calls on the device (fwd_batch, mul_accumulate_batch,
    inv_batch: src/prime64.rs:794, :1085-1128, :872) -- kernels that share nothing with the fused chain but the
    butterflies;
  * on sampled elements (first, last, both sides of every grid-round boundary) with the oracle.
Bit-exact."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

P62, P63, SOLINAS, PM64B = 4611686018427322369, 9223372036853661697, 18446744069414584321, 18446744073707716609
FP50, FP51 = 1125899904679937, 2251799813554177
P30, P31, P32 = 1062862849, 2147352577, 4293918721


def _cus():
    import torch
    return torch.cuda.get_device_properties(0).multi_processor_count


def _dev(a):
    import torch
    return torch.from_numpy(a.view(np.int64 if a.dtype == np.uint64 else np.int32).copy()).cuda()


def _host(t, dtype):
    return t.cpu().numpy().view(dtype)


def _add_mod(a, b, p):
    """canonical a + b mod p on unsigned numpy arrays (any p below 2^bits)"""
    s = a + b
    wrap = (s < a) | (s >= a.dtype.type(p))
    return np.where(wrap, s - a.dtype.type(p), s)


def _separate_calls(plan, dterms, dkey, n, J, O, batch):
    """the chain as the reference's caller writes it, on whole batches: device tensors in, host array out"""
    import torch
    t = dterms.clone()
    plan.fwd_batch(t)
    t3, k3 = t.view(batch, J, n), dkey.view(J, O, n)
    out = torch.empty(batch * O * n, dtype=dterms.dtype, device="cuda")
    for o in range(O):
        acc = torch.zeros(batch * n, dtype=dterms.dtype, device="cuda")
        for j in range(J):
            lhs = t3[:, j, :].contiguous().view(-1)
            rhs = k3[j, o].repeat(batch)
            plan.mul_accumulate_batch(acc, lhs, rhs)
        plan.inv_batch(acc)
        out.view(batch, O, n)[:, o, :] = acc.view(batch, n)
    return out


def _oracle_elements(oplan, terms, key, n, J, O, which):
    want = {}
    kf = key.reshape(J, O, n)
    for b in which:
        acc = [np.zeros(n, dtype=terms.dtype) for _ in range(O)]
        for j in range(J):
            t = terms[(b * J + j) * n:(b * J + j + 1) * n].copy()
            oplan.fwd(t)
            for o in range(O):
                oplan.mul_accumulate(acc[o], t, np.ascontiguousarray(kf[j, o]))
        for o in range(O):
            oplan.inv(acc[o])
        want[b] = np.concatenate(acc)
    return want


# elements per round of the persistent grid, as csrc/ntt_ext_inst.inc launches it on a device with `cus` compute units
def _round_wp(cus):   # ext_kernel_wp: CUs x PER_CU workgroups x PPB elements -- 256 x 2 x 4 or 256 x 1 x 8 at most
    return cus * 8


def _round_blk(cus, n, O):   # ext_kernel_blk: one element per workgroup, PER_CU = 3 / 2 (n = 4096), 1 above
    return cus * ((3 if O <= 2 else 2) if n == 4096 else 1)


def _round_blk32(cus, n):    # ext_kernel_blk on 32-bit words (one output): two 512-thread workgroups per CU at n = 16384, one at 32768
    return cus * (2 if n == 16384 else 1)


CASES = [
    # bits, n, p, J, O, kernel
    (64, 1024, P62, 2, 1, "wp"),
    (64, 1024, P62, 2, 2, "wp"),
    (64, 1024, P62, 2, 4, "wp"),       # four 64-bit accumulator tiles: the shape without the next-term prefetch
    (64, 1024, FP50, 2, 2, "wp"),      # double-precision class
    (64, 1024, SOLINAS, 2, 2, "wp"),   # 2^64 - c
    (64, 2048, P62, 2, 2, "wp"),       # two wavefronts per element: raw s_barrier hand-over
    (64, 2048, P63, 2, 4, "wp"),
    (32, 1024, P30, 2, 1, "wp"),
    (32, 1024, P30, 2, 4, "wp"),
    (32, 1024, P32, 2, 2, "wp"),       # p >= 2^31 on doubles
    (32, 4096, P30, 2, 2, "wp"),
    (32, 4096, P31, 2, 4, "wp"),       # 16-coefficient schedule family
    (64, 4096, P62, 2, 2, "blk"),
    (64, 4096, FP50, 2, 4, "blk"),
    (64, 4096, PM64B, 2, 1, "blk"),
    (64, 8192, P62, 2, 2, "blk"),
    (64, 8192, SOLINAS, 2, 1, "blk"),
    (64, 16384, P62, 2, 2, "blk"),
    (64, 16384, FP51, 2, 1, "blk"),
    (64, 16384, P62, 2, 4, "blk"),     # four outputs at this size: TWO launches of the two-output kernel over the same batch
    (64, 16384, SOLINAS, 2, 3, "blk"),
    # 32-bit words on the wave-block walk (round 4: one output; 30-bit n = 16384, p >= 2^31 n = 16384 / 32768)
    (32, 16384, P30, 2, 1, "blk32"),
    (32, 16384, P32, 2, 1, "blk32"),
    (32, 32768, P32, 2, 1, "blk32"),
]


@pytest.mark.parametrize("idx", range(len(CASES)))
def test_gpu_chain_kernels_walk_the_batch(oracle, idx):
    bits, n, p, J, O, kernel = CASES[idx]
    run_chain_case(oracle, bits, n, p, J, O, kernel, idx % 2 == 1, idx)


def run_chain_case(oracle, bits, n, p, J, O, kernel, accumulate, idx):
    """one multi-trip case (also driven with random primes / shapes by tools/soak_random.py chain)"""
    import torch
    from concrete_ntt_amd import prime32, prime64
    mod = prime64 if bits == 64 else prime32
    dt = np.uint64 if bits == 64 else np.uint32
    plan, oplan = mod.Plan.try_new(n, p), oracle.Plan.try_new(n, p, bits)
    assert plan is not None and oplan is not None
    cus = _cus()
    per_round = _round_wp(cus) if kernel == "wp" else _round_blk32(cus, n) if kernel == "blk32" else _round_blk(cus, n, O)
    batch = 2 * per_round + per_round // 3 + 5   # a third, partial trip with a ragged last tile
    terms = oracle.fill_uniform(batch * J * n, p, 1000 + idx, bits)
    key = oracle.fill_uniform(J * O * n, p, 2000 + idx, bits)
    init = oracle.fill_uniform(batch * O * n, p, 3000 + idx, bits)
    dterms, dkey = _dev(terms), _dev(key)
    dout = _dev(init) if accumulate else torch.zeros(batch * O * n, dtype=dterms.dtype, device="cuda")
    plan.external_product_batch(dout, dterms, dkey, J, O, accumulate)
    got = _host(dout, dt)
    del dout
    # every element against the separate calls
    sep = _host(_separate_calls(plan, dterms, dkey, n, J, O, batch), dt)
    want = _add_mod(init, sep, p) if accumulate else sep
    bad = np.nonzero((got != want).reshape(batch, -1).any(axis=1))[0]
    assert bad.size == 0, ("elements differing from the separate calls", bad[:8], batch, per_round)
    # sampled elements against the oracle
    which = sorted({0, 1, batch - 1, batch - 2} | {r * per_round + d for r in (1, 2) for d in (-1, 0, 1)}
                   | {r * (per_round // 2) + d for r in (1, 2, 3, 4) for d in (-1, 0)})
    which = [b for b in which if 0 <= b < batch]
    for b, w in _oracle_elements(oplan, terms, key, n, J, O, which).items():
        g = got[b * O * n:(b + 1) * O * n]
        w = _add_mod(init[b * O * n:(b + 1) * O * n], w, p) if accumulate else w
        assert np.array_equal(g, w), ("element differs from the oracle", b, batch, per_round)
    assert np.array_equal(_host(dterms, dt), terms) and np.array_equal(_host(dkey, dt), key)   # inputs untouched


@pytest.mark.parametrize("shape,accumulate", [("u32x2", False), ("u30x2", True), ("u32x2_u64x1", False)])
def test_gpu_product_chain_walks_the_batch(oracle, shape, accumulate):
    """cntt_product_external_product_batch (split -> one fused chain per prime plane -> Garner) over three trips of the
    chain kernels' grids: every element against the plan's own separate batched calls (fwd_batch, mul_accumulate_batch
    per (j, o), inv_batch), sampled elements against the oracle's product::Plan (src/product.rs:273, :935, :360)."""
    import torch
    from concrete_ntt_amd import product
    from test_product import _ref_primes
    n, J, O = 1024, 2, 2
    primes = sorted(_ref_primes(oracle, n, shape))
    big = 1
    for q in primes:
        big *= q
    plan, oplan = product.Plan.try_new(n, big, primes), oracle.Product.try_new(n, big, primes)
    per_round = _round_wp(_cus())
    batch = 2 * per_round + per_round // 3 + 5
    n32 = sum(q < 2**32 for q in primes)
    dl = plan.ntt_domain_len()
    terms = oracle.fill_uniform(batch * J * n, big, 71, 64)
    init = oracle.fill_uniform(batch * O * n, big, 72, 64)
    planes = [oracle.fill_uniform(J * O * n, q, 80 + i, 64) for i, q in enumerate(primes)]
    key32 = np.concatenate([pl_.astype(np.uint32) for pl_ in planes[:n32]]) if n32 else np.zeros(0, dtype=np.uint32)
    key = np.concatenate([key32.view(np.uint64)] + [pl_ for pl_ in planes[n32:]])
    imode = product.InvMode.Accumulate if accumulate else product.InvMode.Replace
    dterms, dkey = _dev(terms), _dev(key)
    dout = _dev(init) if accumulate else torch.zeros(batch * O * n, dtype=torch.int64, device="cuda")
    plan.external_product_batch(dout, dterms, dkey, J, O, product.FwdMode.Generic, imode)
    got = _host(dout, np.uint64)
    # the same step as separate batched calls of the same plan (plane-major NTT domain: plane k of a batch of B
    # polynomials holds B * n residues)
    tdom = torch.zeros(batch * J * dl, dtype=torch.int64, device="cuda")
    plan.fwd_batch(tdom, dterms, product.FwdMode.Generic)

    def planes_of(dom, count):   # list of (tensor view [count, n], is32)
        out, off = [], 0
        w32 = dom.view(torch.int32)
        for k in range(n32):
            out.append(w32[k * count * n:(k + 1) * count * n].view(count, n))
        off = n32 * count * n // 2
        for k in range(len(primes) - n32):
            out.append(dom[off + k * count * n: off + (k + 1) * count * n].view(count, n))
        return out

    tpl, kpl = planes_of(tdom, batch * J), planes_of(dkey, J * O)
    sep = torch.empty(batch * O * n, dtype=torch.int64, device="cuda")
    for o in range(O):
        acc = torch.zeros(batch * dl, dtype=torch.int64, device="cuda")
        apl = planes_of(acc, batch)
        for j in range(J):
            lhs = torch.zeros(batch * dl, dtype=torch.int64, device="cuda")
            rhs = torch.zeros(batch * dl, dtype=torch.int64, device="cuda")
            for dst, src in zip(planes_of(lhs, batch), tpl):
                dst.copy_(src.view(batch, J, n)[:, j, :])
            for dst, src in zip(planes_of(rhs, batch), kpl):
                dst.copy_(src[j * O + o].expand(batch, n))
            plan.mul_accumulate_batch(acc, lhs, rhs)
        del apl
        std = dout.new_zeros(batch * n) if not accumulate else _dev(init).view(batch, O, n)[:, o, :].contiguous().view(-1)
        plan.inv_batch(std, acc, imode)
        sep.view(batch, O, n)[:, o, :] = std.view(batch, n)
    want = _host(sep, np.uint64)
    bad = np.nonzero((got != want).reshape(batch, -1).any(axis=1))[0]
    assert bad.size == 0, ("elements differing from the separate calls", bad[:8], batch, per_round)

    # sampled elements against the oracle
    def key_poly(j, o):
        i = j * O + o
        parts = []
        if n32:
            parts.append(np.concatenate([planes[k][i * n:(i + 1) * n].astype(np.uint32) for k in range(n32)]).view(np.uint64))
        parts += [planes[k][i * n:(i + 1) * n] for k in range(n32, len(primes))]
        return np.concatenate(parts)

    for b in (0, per_round - 1, per_round, 2 * per_round - 1, 2 * per_round, batch - 1):
        acc = [np.zeros(dl, dtype=np.uint64) for _ in range(O)]
        for j in range(J):
            t = np.zeros(dl, dtype=np.uint64)
            oplan.fwd(t, terms[(b * J + j) * n:(b * J + j + 1) * n].copy(), None)
            for o in range(O):
                oplan.mul_accumulate(acc[o], t, key_poly(j, o))
        for o in range(O):
            r = init[(b * O + o) * n:(b * O + o + 1) * n].copy() if accumulate else np.zeros(n, dtype=np.uint64)
            oplan.inv(r, acc[o], accumulate)
            assert np.array_equal(got[(b * O + o) * n:(b * O + o + 1) * n], r), ("element differs from the oracle", b, o)
