// product::Plan with two 32-bit primes -- the shape of the reference's own fast path (src/product.rs:295-335 forward,
// :419-789 inverse) -- as ONE kernel per direction: the residue split (FwdMode::Generic `%` or the FwdMode::Bounded
// select) in the load of the forward transforms, the Garner recombination (InvMode::Replace / Accumulate) in the store
// of the inverse transforms (K5 as prologue of K1, K6 as epilogue of K2: SURVEY 2.1).  Both primes must share an
// arithmetic class; twiddles come from the per-prime tables in L2; plane-major batched layout (cntt.h).
// Like the reference, inv leaves the inverse-transformed residues in the ntt buffer.
#pragma once
#include "aux_kernels.hpp"
#include "ntt_kernel.hpp"

namespace cntt {

struct ProductFusedTables {
    const TwPair<uint32_t> *twf[2], *twi[2];
    ModParams<uint32_t> P[2];
};

// MODE 0: FwdMode::Generic; 1: FwdMode::Bounded (host has checked bound < p0, p1)
template <int LOGN, int CLS, int BLK>
__global__ __launch_bounds__(BLK, 2) void product_fwd2_kernel(uint32_t *__restrict__ res32, const uint64_t *__restrict__ standard,
                                                              const ProductFusedTables F, const ProductArgs A, uint32_t batch,
                                                              uint32_t bounded) {
    using Wf = NttWp<uint32_t, LOGN, false, CLS, BLK, 3>;   // family 3: 16 coefficients per thread, padded exchange layout (round 4)
    constexpr int E = Wf::E, TPP = Wf::TPP, NPASS = Wf::NPASS, PPB = BLK / TPP;
    constexpr uint32_t FULL = Wf::FULL, RM0 = Wf::S::RMASK[0], RML = Wf::S::RMASK[NPASS - 1];
    __shared__ __attribute__((aligned(16))) uint32_t lds_all[(size_t)PPB * Wf::B::LDS_WORDS_1];
    const uint32_t tid = threadIdx.x & (TPP - 1), pl = threadIdx.x / TPP;
    uint32_t *lds = lds_all + (size_t)pl * Wf::B::LDS_WORDS_1;
    const uint32_t sub = blockIdx.x * PPB + pl;
    const uint32_t subc = sub < batch ? sub : batch - 1;
    const uint64_t *sp = standard + ((size_t)subc << LOGN);
    const uint32_t ebase0 = pdep<FULL & ~RM0>(tid), ebaseL = pdep<FULL & ~RML>(tid);
    const size_t plane = (size_t)batch << LOGN;  // residues per plane
    const uint32_t p0 = (uint32_t)A.prime[0], p1 = (uint32_t)A.prime[1], pu = (uint32_t)A.modulus;
    const uint64_t half = A.modulus / 2;
    static_for<0, 2>([&](auto ic) {
        constexpr int i = ic.value;
        uint32_t a[E];
#pragma unroll
        for (int j = 0; j < E; ++j) {
            const uint64_t s = sp[ebase0 | cdep((uint32_t)j, RM0)];
            if (bounded) {  // src/product.rs:303-322, computed exactly as the reference does
                const uint32_t sx = (uint32_t)s, c = pu - sx, pi = i == 0 ? p0 : p1;
                a[j] = s < half ? sx : pi - c;
            } else {
                a[j] = (uint32_t)barrett_rem(s, A.prime[i], A.barrett[i]);
            }
            a[j] = Bfly<uint32_t, CLS>::load_fix(a[j], F.P[i]);  // canonical word -> the class's register form
        }
        Wf::template pass<0, false, false>(a, lds, tid, F.twf[i], nullptr, F.P[i]);
        if (sub < batch) Wf::B::template scatter<RML>(a, res32 + (size_t)i * plane + ((size_t)sub << LOGN), ebaseL, false);
        Wf::wsync();
    });
}

// ACC 0: InvMode::Replace; 1: InvMode::Accumulate (add_mod_u64 with the reference's overflow arm, src/product.rs:85-92)
template <int LOGN, int CLS, int BLK>
__global__ __launch_bounds__(BLK, 2) void product_inv2_kernel(uint64_t *__restrict__ standard, uint32_t *__restrict__ res32,
                                                              const ProductFusedTables F, const ProductArgs A, uint32_t batch,
                                                              uint32_t accumulate) {
    using Wi = NttWp<uint32_t, LOGN, true, CLS, BLK, 3>;
    constexpr int E = Wi::E, TPP = Wi::TPP, NPASS = Wi::NPASS, PPB = BLK / TPP;
    constexpr uint32_t FULL = Wi::FULL, RM0 = Wi::S::RMASK[0], RML = Wi::S::RMASK[NPASS - 1];
    __shared__ __attribute__((aligned(16))) uint32_t lds_all[(size_t)PPB * Wi::B::LDS_WORDS_1];
    const uint32_t tid = threadIdx.x & (TPP - 1), pl = threadIdx.x / TPP;
    uint32_t *lds = lds_all + (size_t)pl * Wi::B::LDS_WORDS_1;
    const uint32_t sub = blockIdx.x * PPB + pl;
    const uint32_t subc = sub < batch ? sub : batch - 1;
    const uint32_t ebase0 = pdep<FULL & ~RM0>(tid), ebaseL = pdep<FULL & ~RML>(tid);
    const size_t plane = (size_t)batch << LOGN;
    uint32_t res[2][E];
    static_for<0, 2>([&](auto ic) {
        constexpr int i = ic.value;
        uint32_t *rp = res32 + (size_t)i * plane + ((size_t)subc << LOGN);
        uint32_t a[E];
        Wi::B::template gather<RM0>(a, (const uint32_t *)rp, ebase0, false);
#pragma unroll
        for (int j = 0; j < E; ++j) a[j] = Bfly<uint32_t, CLS>::load_fix(a[j], F.P[i]);
        Wi::template pass<0, false, false>(a, lds, tid, F.twi[i], nullptr, F.P[i]);
        if (sub < batch) Wi::B::template scatter<RML>(a, rp, ebaseL, false);  // inv(ntt) stays in the ntt buffer: src/product.rs:368-373
        Wi::wsync();
#pragma unroll
        for (int j = 0; j < E; ++j) res[i][j] = a[j];
    });
    if (sub < batch) {
        uint64_t *op = standard + ((size_t)sub << LOGN);
#pragma unroll
        for (int j = 0; j < E; ++j) {
            const uint64_t u[2] = {res[0][j], res[1][j]};
            uint64_t out = garner<2>(u, A);
            const uint32_t e = ebaseL | cdep((uint32_t)j, RML);
            if (accumulate) out = add_mod_u64(A.modulus, op[e], out);
            op[e] = out;
        }
    }
}

}  // namespace cntt
