"""Batches past 2^32 ELEMENTS: the batch counters of the kernels are 32-bit kernel arguments and several address forms
are "workgroup-uniform 64-bit base + 32-bit lane offset" (DESIGN 3), so one transform and the pointwise kernels are run ONCE on a u32
batch of 16 GiB + one polynomial -- element indices beyond 2^32, byte offsets beyond 2^34 -- on the two kernel families that serve
32-bit words (persistent LDS-resident walk at n = 1024, wave-block walk at n = 32768):
  * the FIRST and the LAST polynomial (the one whose first element index is exactly 2^32) against the oracle, word for word;
  * every polynomial through size-independent properties: inv(fwd(x)) normalised == x, and mul_assign_normalize by the all-ones
    NTT-domain polynomial == normalize  (reference: src/prime32.rs:709-808 fwd / inv, :812-927 pointwise; its own test
    src/prime32.rs:1006-1060 checks the same identities on one polynomial)."""
import numpy as np
import pytest

from concrete_ntt_amd import prime32
import concrete_ntt_amd as cntt

pytestmark = pytest.mark.gpu

P30 = 1062862849


def _equal(torch, a, b, chunk=1 << 28):
    for i in range(0, a.numel(), chunk):
        if not torch.equal(a[i:i + chunk], b[i:i + chunk]):
            return False
    return True


@pytest.mark.parametrize("n", [1024, 32768])
def test_u32_batch_of_16_gib_plus_one_polynomial(oracle, n):
    import torch
    assert torch.cuda.is_available(), "these tests need the MI355X"
    batch = (1 << 32) // n + 1
    count = batch * n
    assert count > (1 << 32) and (batch - 1) * n == 1 << 32
    free, _ = torch.cuda.mem_get_info()
    if free < 3.3 * count * 4:
        pytest.skip("needs three 16 GiB buffers; %.0f GiB free" % (free / 2**30))
    plan, ref = prime32.Plan.try_new(n, P30), oracle.Plan.try_new(n, P30, 32)
    x = torch.empty(count, dtype=torch.int32, device="cuda")
    cntt.fill_uniform(x, P30, 0x5EED0B16)
    torch.cuda.synchronize()
    ends = {i: x[i * n:(i + 1) * n].cpu().numpy().view(np.uint32).copy() for i in (0, 1, batch - 2, batch - 1)}
    for v in ends.values():
        assert int(v.max()) < P30 and int(v.max()) > 0      # the fill itself reached past 2^32 elements
    y = x.clone()
    # ---- fwd: both ends against the oracle -------------------------------------------------------------------------
    plan.fwd_batch(y)
    torch.cuda.synchronize()
    for i, v in ends.items():
        want = v.copy()
        ref.fwd(want)
        got = y[i * n:(i + 1) * n].cpu().numpy().view(np.uint32)
        assert np.array_equal(got, want), ("fwd", n, i)
    # ---- mul_assign_normalize past 2^32 elements: by the all-ones polynomial it is normalize -----------------------
    ones = torch.ones(count, dtype=torch.int32, device="cuda")
    z = y.clone()
    plan.mul_assign_normalize_batch(z, ones)
    del ones
    w = y.clone()
    plan.normalize_batch(w)
    torch.cuda.synchronize()
    for i in (0, batch - 1):
        want = ends[i].copy()
        ref.fwd(want)
        ref.normalize(want)
        assert np.array_equal(w[i * n:(i + 1) * n].cpu().numpy().view(np.uint32), want), ("normalize", n, i)
    assert _equal(torch, z, w), "mul_assign_normalize(., 1) != normalize somewhere in the batch"
    del z
    # ---- inv: the whole batch comes back (w = fwd(x) / n, so inv(w) == x), both ends against the oracle too ---------
    plan.inv_batch(w)
    torch.cuda.synchronize()
    assert _equal(torch, w, x), "inv(normalize(fwd(x))) != x somewhere in the batch"
    plan.inv_batch(y)                          # the unnormalised inverse of the last polynomial, against the oracle
    torch.cuda.synchronize()
    for i in (0, batch - 1):
        want = ends[i].copy()
        ref.fwd(want)
        ref.inv(want)
        assert np.array_equal(y[i * n:(i + 1) * n].cpu().numpy().view(np.uint32), want), ("inv", n, i)
