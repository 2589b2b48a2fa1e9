"""BASELINE.json configs at their FULL batch sizes (the slice tests of test_gpu_parity.py cover every polynomial of a
small batch): C3 native64 N=4096 x 16384, C5 native_binary64 N=2048 x 65536, C4's whole per-GPU shard prime64 N=16384 x
131072 (16 GiB in place).  A batch this large cannot go through the CPU oracle in test time, so each case checks
  * a sample of polynomials (first, last, the workgroup / tile boundaries of the kernels, a few in between) bit for bit
    against the oracle, regenerated on the host from the same splitmix64 stream the device fill uses, and
  * EVERY polynomial against a second, independent device path (the composed split -> per-prime transforms ->
    pointwise -> CRT pipeline for the fused native kernels; the inverse transform + normalize round trip for C4)."""
import numpy as np
import pytest

import concrete_ntt_amd as cntt
from concrete_ntt_amd import native64, native_binary64, prime64

pytestmark = pytest.mark.gpu

P62 = 4611686018427322369


def _torch():
    import torch
    assert torch.cuda.is_available(), "these tests need the MI355X"
    return torch


def to_host(t):
    return t.cpu().numpy().view(np.uint64)


@pytest.mark.parametrize("kind,cls,n,batch,seed", [("native64_plan32", native64.Plan32, 4096, 16384, 0x5EED0003),
                                                   ("native_binary64_plan32", native_binary64.Plan32, 2048, 65536, 0x5EED0005)])
def test_native_polymul_full_batch(oracle, kind, cls, n, batch, seed):
    torch = _torch()
    plan, ref = cls.try_new(n), oracle.Native(kind, n)
    lhs = torch.empty(batch * n, dtype=torch.int64, device="cuda")
    rhs = torch.empty_like(lhs)
    cntt.fill_uniform(lhs, 0, seed)
    cntt.fill_uniform(rhs, 0, seed + (1 << 40))
    if cls.BINARY:
        rhs &= 1
    prod = torch.empty_like(lhs)
    plan.negacyclic_polymul_batch(prod, lhs, rhs)          # the whole-product kernel (csrc/native_fused.hpp)
    # (1) sampled polynomials against the oracle
    sample = sorted({0, 1, 2, 3, 255, 256, 257, batch // 2 - 1, batch // 2, batch - 257, batch - 2, batch - 1,
                     12345 % batch, 7777 % batch})
    for s in sample:
        hl = oracle.fill_uniform(n, 0, seed + s * n, 64)
        hr = oracle.fill_uniform(n, 0, seed + (1 << 40) + s * n, 64)
        if cls.BINARY:
            hr &= np.uint64(1)
        want = np.zeros(n, dtype=np.uint64)
        ref.negacyclic_polymul(want, hl, hr)
        assert np.array_equal(to_host(prod[s * n:(s + 1) * n]), want), (kind, s)
    # (2) every polynomial against the composed pipeline: fwd / fwd_binary -> per-prime mul_assign_normalize -> inv
    res_dt = torch.int32
    rl = [torch.empty(batch * n, dtype=res_dt, device="cuda") for _ in range(cls.NPRIMES)]
    rr = [torch.empty(batch * n, dtype=res_dt, device="cuda") for _ in range(cls.NPRIMES)]
    plan.fwd_batch(lhs, rl)
    plan.fwd_batch(rhs, rr, binary=cls.BINARY)
    for i in range(cls.NPRIMES):
        plan.ntt(i).mul_assign_normalize_batch(rl[i], rr[i])
    composed = torch.empty_like(lhs)
    plan.inv_batch(composed, rl)
    assert torch.equal(prod, composed), kind


def test_c4_whole_shard(oracle):
    """prime64 N=16384, 131072 polynomials = 16 GiB in place (one GPU's shard of the 2^20-polynomial C4 batch)."""
    torch = _torch()
    n, batch, seed = 16384, 131072, 0x5EED0004
    free, _ = torch.cuda.mem_get_info()
    if free < 40 * 2**30:
        pytest.skip("needs 2 x 16 GiB of free HBM")
    plan, ref = prime64.Plan.try_new(n, P62), oracle.Plan.try_new(n, P62, 64)
    a = torch.empty(batch * n, dtype=torch.int64, device="cuda")
    cntt.fill_uniform(a, P62, seed)
    a0 = a.clone()
    plan.fwd_batch(a)
    for s in (0, 1, 255, 256, batch // 2, batch - 257, batch - 1):
        want = oracle.fill_uniform(n, P62, seed + s * n, 64)
        ref.fwd(want)
        assert np.array_equal(to_host(a[s * n:(s + 1) * n]), want), s
    plan.inv_batch(a)
    plan.normalize_batch(a)          # inv(fwd(x)) = N x  (src/prime64.rs:866-871); normalize multiplies by 1/N
    assert torch.equal(a, a0)
