"""Deterministic worst cases of the accumulating-CRT whole-product kernel (native_polymul_kernel_acc, csrc/native_fused.hpp; VERDICT
round 4).  The kernel decides which multiple of M = P_0 ... P_{k-1} to subtract by ROUNDING a 27-bit fixed-point sum of gamma_i / P_i;
the exact sum is k + c / M for the exact integer coefficient c of the product, so the rounding is safe while |c| / M plus the
fixed-point error stays below 1/2 (2^-5.9 + k 2^-26).  Uniform random words never come near the bound on |c|; these
patterns reach it:
    all-ones (.) all-ones          c_j = A^2 (2j + 2 - n): the largest positive value n A^2 at j = n - 1, -(n - 2) A^2 at j = 0
    all-ones (.) (0, A, A, ... A)  c_0 = -(n - 1) A^2: the most negative coefficient a negacyclic product can have
    all-ones (.) spike at 0 / n-1  one term per coefficient, every sign pattern of the wrap-around
    all-ones (.) alternating 0 / A half the terms, alternating signs across the wrap
(binary kinds: rhs words in {0, 1} -- src/native_binary64.rs:379-385 takes them `as u32`).  For each of the six Plan32 kinds at the
largest n of each kernel shape: against the oracle's negacyclic_polymul (src/native64.rs:1042-1069 and the per-kind CRTs), and against
the parked-tile kernel (`native_acc` switch off, include/cntt.h testing-only switchboard) -- two device paths with different CRTs."""
import numpy as np
import pytest

import concrete_ntt_amd as cntt
from concrete_ntt_amd import native32, native64, native128, native_binary32, native_binary64, native_binary128

pytestmark = pytest.mark.gpu

KINDS = {"native32_plan32": native32.Plan32, "native64_plan32": native64.Plan32, "native128_plan32": native128.Plan32,
         "native_binary32_plan32": native_binary32.Plan32, "native_binary64_plan32": native_binary64.Plan32,
         "native_binary128_plan32": native_binary128.Plan32}


def _patterns(n, word, dtype, binary):
    """[(name, lhs, rhs)] as flat word arrays (128-bit words: two u64, low first)."""
    wpp = n * (2 if word == 16 else 1)
    full = np.iinfo(dtype).max
    ones = np.full(wpp, full, dtype=dtype)

    def rhs_of(mask):               # mask: bool per coefficient -> that coefficient is "big" (all-ones word / binary 1)
        r = np.zeros(wpp, dtype=dtype)
        if word == 16:
            r[0::2][mask] = 1 if binary else full
            if not binary:
                r[1::2][mask] = full
        else:
            r[mask] = 1 if binary else full
        return r
    idx = np.arange(n)
    out = [("ones x ones", ones, rhs_of(idx >= 0)),
           ("ones x (0, A, ..., A)", ones, rhs_of(idx >= 1)),
           ("ones x spike at 0", ones, rhs_of(idx == 0)),
           ("ones x spike at n-1", ones, rhs_of(idx == n - 1)),
           ("ones x alternating", ones, rhs_of(idx % 2 == 1)),
           ("ones x upper half", ones, rhs_of(idx >= n // 2))]
    if not binary:   # the same with the roles swapped where the operands differ in kind
        out.append(("(0, A, ..., A) x ones", rhs_of(idx >= 1), ones))
    return out


@pytest.mark.parametrize("n", [4096, 16384, 32768])
@pytest.mark.parametrize("kind", sorted(KINDS))
def test_accumulating_crt_at_the_bound_of_the_coefficients(oracle, kind, n):
    import torch
    assert torch.cuda.is_available(), "these tests need the MI355X"
    cls = KINDS[kind]
    plan, ref = cls.try_new(n), oracle.Native(kind, n)
    dtype = np.uint32 if ref.word == 4 else np.uint64
    pats = _patterns(n, ref.word, dtype, cls.BINARY)
    reps = 3                                     # 3 x 7 products: more than one workgroup's share at every shape, ragged tail
    lhs = np.concatenate([p[1] for p in pats] * reps)
    rhs = np.concatenate([p[2] for p in pats] * reps)
    batch = len(pats) * reps
    wpp = lhs.size // batch
    want = np.zeros_like(lhs)
    ref.negacyclic_polymul_batch(want, lhs, rhs, batch, 8)
    sdt = np.int64 if dtype == np.uint64 else np.int32

    def run():
        dp = torch.zeros(lhs.size, dtype=torch.int64 if dtype == np.uint64 else torch.int32, device="cuda")
        plan.negacyclic_polymul_batch(dp, torch.from_numpy(lhs.view(sdt)).cuda(), torch.from_numpy(rhs.view(sdt)).cuda())
        return dp.cpu().numpy().view(dtype)
    got = run()
    for b in range(batch):
        sl = slice(b * wpp, (b + 1) * wpp)
        bad = np.nonzero(got[sl] != want[sl])[0]
        assert bad.size == 0, "%s n=%d pattern %r (product %d): %d words differ from the oracle, first at %d" % (
            kind, n, pats[b % len(pats)][0], b, bad.size, bad[0])
    with cntt.debug_switches(native_acc=0):      # the parked-tile kernels: mixed-radix CRT as the reference writes it
        parked = run()
    assert np.array_equal(parked, want), (kind, n, "parked-tile kernel")
