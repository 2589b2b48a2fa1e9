// Whole negacyclic_polymul of the native / native_binary Plan32 kinds in ONE kernel: for every
// prime  split(lhs), split(rhs) -> two forward transforms -> pointwise product -> inverse transform (1/N folded into
// its last stage)  with the residues kept in registers, then the mixed-radix CRT on the register tiles and one store
// of the product word.  HBM traffic per product: lhs + rhs read ONCE (kept in registers across the primes), prod
// written (the fused lower bound of SURVEY 8(d)); the unfused pipeline moves the 2k residue arrays through HBM
// several times.
// Values are those of src/native64.rs:1042-1069 (and siblings): same split (% P_i), same transforms, same digits.
// Twiddles come from the per-prime tables in global memory (L2): an LDS image per prime would not fit.
#pragma once
#include <type_traits>
#include "aux_kernels.hpp"
#include "ntt_kernel.hpp"

namespace cntt {

// digit structure of the reference plans handled here (host.hip NATIVE_KINDS; kind numbers of cntt_native_kind_t)
template <int KIND> struct NativeShape;
template <> struct NativeShape<0> {  // native32::Plan32: three single-prime digits, u32 words
    using W = uint32_t;
    static constexpr int KP = 3, NG = 3;
    static constexpr uint32_t PAIRS = 0u;
    static constexpr bool BINARY = false;
    static constexpr int ga(int g) { return g; }
    static constexpr int gb(int) { return -1; }
};
template <> struct NativeShape<1> {  // native64::Plan32: digits P0, (P1,P2), (P3,P4), u64 words
    using W = uint64_t;
    static constexpr int KP = 5, NG = 3;
    static constexpr uint32_t PAIRS = 0b110u;
    static constexpr bool BINARY = false;
    static constexpr int ga(int g) { return g == 0 ? 0 : g == 1 ? 1 : 3; }
    static constexpr int gb(int g) { return g == 0 ? -1 : g == 1 ? 2 : 4; }
};
template <> struct NativeShape<3> {  // native_binary32::Plan32
    using W = uint32_t;
    static constexpr int KP = 2, NG = 2;
    static constexpr uint32_t PAIRS = 0u;
    static constexpr bool BINARY = true;
    static constexpr int ga(int g) { return g; }
    static constexpr int gb(int) { return -1; }
};
template <> struct NativeShape<4> {  // native_binary64::Plan32
    using W = uint64_t;
    static constexpr int KP = 3, NG = 3;
    static constexpr uint32_t PAIRS = 0u;
    static constexpr bool BINARY = true;
    static constexpr int ga(int g) { return g; }
    static constexpr int gb(int) { return -1; }
};

struct alignas(16) Word128 {  // u128 as Rust lays it out on x86-64: 16-byte little-endian (lo, hi)
    uint64_t lo, hi;
};
template <> struct NativeShape<5> {  // native_binary128::Plan32: digits P0, (P1,P2), (P3,P4), u128 words
    using W = Word128;
    static constexpr int KP = 5, NG = 3;
    static constexpr uint32_t PAIRS = 0b110u;
    static constexpr bool BINARY = true;
    static constexpr int ga(int g) { return g == 0 ? 0 : g == 1 ? 1 : 3; }
    static constexpr int gb(int g) { return g == 0 ? -1 : g == 1 ? 2 : 4; }
};

template <int KP> struct FusedTables {
    const TwPair<uint32_t> *twf[KP], *twi[KP];
    ModParams<uint32_t> P[KP];
};

// value % P_k for a u32 / u64 word, canonical (split_kernel's 30-bit-prime branch)
template <class W> __device__ __forceinline__ uint32_t split30(W w, const SplitArgs &A, int k) {
    const uint32_t p = (uint32_t)A.prime[k];
    if constexpr (sizeof(W) == 16) {
        uint32_t acc = (uint32_t)(w.hi >> 32);
        acc = fold32(acc, (uint32_t)w.hi, p, A.c[k], A.c_shoup[k], A.one_shoup[k]);
        acc = fold32(acc, (uint32_t)(w.lo >> 32), p, A.c[k], A.c_shoup[k], A.one_shoup[k]);
        acc = fold32(acc, (uint32_t)w.lo, p, A.c[k], A.c_shoup[k], A.one_shoup[k]);
        return canon4(acc, p);
    } else if constexpr (sizeof(W) == 4) {
        return canon4(red32_lazy((uint32_t)w, p, A.one_shoup[k]), p);
    } else {
        const uint32_t acc = fold32((uint32_t)((uint64_t)w >> 32), (uint32_t)w, p, A.c[k], A.c_shoup[k], A.one_shoup[k]);
        return canon4(acc, p);
    }
}

// crt_kernel's recombination on one coefficient whose KP residues sit in registers (static indices throughout)
template <class SH> __device__ __forceinline__ typename SH::W crt_regs(const uint32_t (&r)[SH::KP], const CrtArgs &A) {
    constexpr int NG = SH::NG;
    uint64_t rg[NG];
#pragma unroll
    for (int g = 0; g < NG; ++g) {
        const uint64_t va = r[SH::ga(g)];
        if (((SH::PAIRS >> g) & 1u) != 0u) {
            const uint64_t pa = A.prime[SH::ga(g)], pb = A.prime[SH::gb(g) < 0 ? 0 : SH::gb(g)];
            const uint64_t mb = r[SH::gb(g) < 0 ? 0 : SH::gb(g)];
            const uint32_t d = (uint32_t)(2 * pb + mb - va);
            const uint64_t vb = shoup_mulmod32(d, (uint32_t)A.pair_inv[g], A.pair_inv_shoup[g], (uint32_t)pb);
            rg[g] = va + vb * pa;
        } else {
            rg[g] = va;
        }
    }
    uint64_t v[NG];
    v[0] = rg[0];
#pragma unroll
    for (int g = 1; g < NG; ++g) {
        if constexpr (SH::PAIRS == 0u) {
            const uint32_t m = (uint32_t)A.M[g];
            uint32_t acc = (uint32_t)v[g - 1];
            acc = umin<uint32_t>(acc, acc - m);
#pragma unroll
            for (int h = g - 2; h >= 0; --h) {
                uint32_t t = shoup_mulmod32(acc, (uint32_t)A.Mmod[g][h], A.Mmod_shoup32[g][h], m);
                uint32_t vh = (uint32_t)v[h];
                vh = umin<uint32_t>(vh, vh - m);
                t += vh;
                acc = umin<uint32_t>(t, t - m);
            }
            const uint32_t d = (uint32_t)rg[g] - acc + m;
            v[g] = shoup_mulmod32(d, (uint32_t)A.inv[g], A.inv_shoup32[g], m);
        } else {
            // the digit moduli of these plans ascend (P0 < P1 P2 < P3 P4), so every digit v[h] < M[h] < M[g] and the
            // group residue rg[g] < M[g] are already canonical modulo M[g]: no reduction (crt_kernel keeps a guarded
            // `%` there; a 64-bit `%` expands to a long division routine, ruinous next to five live residue tiles)
            const uint64_t m = A.M[g];
            uint64_t acc = v[g - 1];
#pragma unroll
            for (int h = g - 2; h >= 0; --h) {
                uint64_t t = shoup_mulmod(acc, A.Mmod[g][h], A.Mmod_shoup[g][h], m);
                t += v[h];
                acc = t >= m ? t - m : t;
            }
            const uint64_t rr = rg[g];
            const uint64_t d = rr >= acc ? rr - acc : rr + m - acc;
            v[g] = shoup_mulmod(d, A.inv[g], A.inv_shoup[g], m);
        }
    }
    const bool sign = v[NG - 1] > (A.M[NG - 1] / 2);
    if constexpr (sizeof(typename SH::W) == 16) {  // recombination modulo 2^128
        u128d pos = {v[0], 0};
#pragma unroll
        for (int g = 1; g < NG; ++g) pos = add128(pos, mul_64x128(v[g], A.prefix_lo[g], A.prefix_hi[g]));
        const u128d full = {A.prefix_lo[NG], A.prefix_hi[NG]};
        const u128d out = sign ? sub128(pos, full) : pos;
        return typename SH::W{out.lo, out.hi};
    } else {  // words of at most 64 bits: the recombination wraps modulo 2^64
        uint64_t pos = v[0];
#pragma unroll
        for (int g = 1; g < NG; ++g) pos += v[g] * A.prefix_lo[g];
        const uint64_t out = sign ? pos - A.prefix_lo[NG] : pos;
        return (typename SH::W)out;
    }
}

// PARK: the first PARK finished residue tiles wait for the CRT in LDS (thread-private slots, no synchronisation)
// instead of registers -- five tiles of a 4096-point product do not fit the 256 VGPRs of two waves per SIMD.
template <int KIND, int LOGN, int BLK, int PARK = 0>
__global__ __launch_bounds__(BLK, 2) void native_polymul_kernel(typename NativeShape<KIND>::W *__restrict__ prod,
                                                             const typename NativeShape<KIND>::W *__restrict__ lhs,
                                                             const typename NativeShape<KIND>::W *__restrict__ rhs,
                                                             const FusedTables<NativeShape<KIND>::KP> F, const SplitArgs S,
                                                             const CrtArgs C, uint32_t batch) {
    using SH = NativeShape<KIND>;
    using W = typename SH::W;
    using Wf = NttWp<uint32_t, LOGN, false, CLS_LAZY, BLK, 1>;
    using Wi = NttWp<uint32_t, LOGN, true, CLS_LAZY, BLK, 1>;
    constexpr int E = Wf::E, TPP = Wf::TPP, NPASS = Wf::NPASS, PPB = BLK / TPP, KP = SH::KP;
    constexpr uint32_t FULL = Wf::FULL, RM0 = Wf::S::RMASK[0];
    static_assert(RM0 == Wi::S::RMASK[NPASS - 1] && Wf::S::RMASK[NPASS - 1] == Wi::S::RMASK[0],
                  "forward and inverse schedules must mirror each other");
    __shared__ __attribute__((aligned(16))) uint32_t lds_all[(size_t)PPB << LOGN];
    __shared__ uint32_t park[PARK > 0 ? PARK : 1][PARK > 0 ? (size_t)BLK * E : 1];
    const uint32_t tid = threadIdx.x & (TPP - 1), pl = threadIdx.x / TPP;
    uint32_t *lds = lds_all + ((size_t)pl << LOGN);
    const uint32_t sub = blockIdx.x * PPB + pl;
    const uint32_t subc = sub < batch ? sub : batch - 1;  // ragged tail: recompute the last polynomial, store nothing
    const W *lp = lhs + ((size_t)subc << LOGN), *rp = rhs + ((size_t)subc << LOGN);
    const uint32_t ebase = pdep<FULL & ~RM0>(tid);
    uint32_t res[KP - PARK][E];
    // Both operands are read ONCE and stay in registers for all KP primes (a re-read per prime came back from HBM
    // more often than not: rocprofv3 FETCH_SIZE showed 2.9x the operand bytes for native64 N=4096 -- profiles/r02).
    W lw[E];
    typename std::conditional<SH::BINARY, uint32_t, W>::type rw[E];
#pragma unroll
    for (int j = 0; j < E; ++j) lw[j] = lp[ebase | cdep((uint32_t)j, RM0)];
#pragma unroll
    for (int j = 0; j < E; ++j) {
        const W w = rp[ebase | cdep((uint32_t)j, RM0)];
        if constexpr (SH::BINARY) {  // `as u32`: src/native_binary64.rs:379-385, src/native_binary128.rs:100-106
            if constexpr (sizeof(W) == 16) rw[j] = (uint32_t)w.lo;
            else rw[j] = (uint32_t)w;
        } else {
            rw[j] = w;
        }
    }
    static_for<0, KP>([&](auto ic) {
        constexpr int i = ic.value;
        uint32_t a[E], b[E];
#pragma unroll
        for (int j = 0; j < E; ++j) a[j] = split30<W>(lw[j], S, i);
        Wf::template pass<0, false, false>(a, lds, tid, F.twf[i], nullptr, F.P[i]);
        Wf::wsync();
#pragma unroll
        for (int j = 0; j < E; ++j) {
            if constexpr (SH::BINARY) b[j] = rw[j];
            else b[j] = split30<W>(rw[j], S, i);
        }
        Wf::template pass<0, false, false>(b, lds, tid, F.twf[i], nullptr, F.P[i]);
#pragma unroll
        for (int j = 0; j < E; ++j) a[j] = mul_for_inv<uint32_t, CLS_LAZY>(a[j], b[j], F.P[i]);
        Wf::wsync();
        Wi::template pass<0, true, false>(a, lds, tid, F.twi[i], nullptr, F.P[i]);
        Wf::wsync();
        if constexpr (i < PARK) {
#pragma unroll
            for (int j = 0; j < E; ++j) park[i][(size_t)j * BLK + threadIdx.x] = a[j];
        } else {
#pragma unroll
            for (int j = 0; j < E; ++j) res[i - PARK][j] = a[j];
        }
    });
    if (sub < batch) {
        W *op = prod + ((size_t)sub << LOGN);
#pragma unroll
        for (int j = 0; j < E; ++j) {
            uint32_t r[KP];
#pragma unroll
            for (int i = 0; i < KP; ++i) r[i] = i < PARK ? park[i < PARK ? i : 0][(size_t)j * BLK + threadIdx.x] : res[i < PARK ? 0 : i - PARK][j];
            op[ebase | cdep((uint32_t)j, RM0)] = crt_regs<SH>(r, C);
        }
    }
}

}  // namespace cntt
