// Whole negacyclic_polymul of the native / native_binary Plan32 kinds in ONE kernel: for every
// prime  split(lhs), split(rhs) -> two forward transforms -> pointwise product -> inverse transform (1/N folded into
// its last stage)  with the residues kept in registers, then the mixed-radix CRT on the residue tiles and one store
// of the product word.  HBM traffic per product at n <= 4096: lhs + rhs read ONCE (kept in registers across the
// primes), prod written (the fused lower bound of SURVEY 8(d)); the unfused pipeline moves the 2k residue arrays
// through HBM several times.  n = 8192 / 16384 re-read the operands per prime and park residue tiles in a small
// cache-resident scratch area (native_polymul_kernel_g).
// Values are those of src/native64.rs:1042-1069 (and siblings): same split (% P_i), same transforms, same digits.
// Twiddles come from the per-prime tables in global memory (L2): an LDS image per prime would not fit.
#pragma once
#include <type_traits>
#include "aux_kernels.hpp"
#include "ntt_kernel.hpp"

namespace cntt {

// digit structure of the reference plans handled here (host.hip NATIVE_KINDS; kind numbers of cntt_native_kind_t)
template <int KIND> struct NativeShape;
struct alignas(16) Word128 {  // u128 as Rust lays it out on x86-64: 16-byte little-endian (lo, hi)
    uint64_t lo, hi;
};
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
template <> struct NativeShape<2> {  // native128::Plan32: five two-prime digits (P0,P1) .. (P8,P9), u128 words
    using W = Word128;
    static constexpr int KP = 10, NG = 5;
    static constexpr uint32_t PAIRS = 0b11111u;
    static constexpr bool BINARY = false;
    static constexpr int ga(int g) { return 2 * g; }
    static constexpr int gb(int g) { return 2 * g + 1; }
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
    static_for<0, NG>([&](auto gc) {
        constexpr int g = gc.value;
        const uint64_t va = r[SH::ga(g)];
        if constexpr (((SH::PAIRS >> g) & 1u) != 0u) {
            const uint64_t pa = A.prime[SH::ga(g)], pb = A.prime[SH::gb(g)];
            const uint64_t mb = r[SH::gb(g)];
            const uint32_t d = (uint32_t)(2 * pb + mb - va);
            const uint64_t vb = shoup_mulmod32(d, (uint32_t)A.pair_inv[g], A.pair_inv_shoup[g], (uint32_t)pb);
            rg[g] = va + vb * pa;
        } else {
            rg[g] = va;
        }
    });
    uint64_t v[NG];
    v[0] = rg[0];
    // (static_for, not `#pragma unroll`: at five digits the nested loops were left rolled and v / rg went to scratch memory)
    static_for<1, NG>([&](auto gc) {
        constexpr int g = gc.value;
        if constexpr (SH::PAIRS == 0u) {
            const uint32_t m = (uint32_t)A.M[g];
            uint32_t acc = (uint32_t)v[g - 1];
            acc = umin<uint32_t>(acc, acc - m);
            static_for<0, g - 1>([&](auto hc) {
                constexpr int h = g - 2 - hc.value;
                uint32_t t = shoup_mulmod32(acc, (uint32_t)A.Mmod[g][h], A.Mmod_shoup32[g][h], m);
                uint32_t vh = (uint32_t)v[h];
                vh = umin<uint32_t>(vh, vh - m);
                t += vh;
                acc = umin<uint32_t>(t, t - m);
            });
            const uint32_t d = (uint32_t)rg[g] - acc + m;
            v[g] = shoup_mulmod32(d, (uint32_t)A.inv[g], A.inv_shoup32[g], m);
        } else {
            // the digit moduli of these plans ascend (P0 < P1 P2 < P3 P4; P0 P1 < P2 P3 < ... for native128), so every digit
            // v[h] < M[h] < M[g] and the group residue rg[g] < M[g] are already canonical modulo M[g]: no reduction
            // (crt_kernel keeps a guarded `%` there; a 64-bit `%` expands to a long division routine, ruinous next to
            // the live residue tiles)
            const uint64_t m = A.M[g];
            uint64_t acc = v[g - 1];
            static_for<0, g - 1>([&](auto hc) {
                constexpr int h = g - 2 - hc.value;
                uint64_t t = shoup_mulmod(acc, A.Mmod[g][h], A.Mmod_shoup[g][h], m);
                t += v[h];
                acc = t >= m ? t - m : t;
            });
            const uint64_t rr = rg[g];
            const uint64_t d = rr >= acc ? rr - acc : rr + m - acc;
            v[g] = shoup_mulmod(d, A.inv[g], A.inv_shoup[g], m);
        }
    });
    const bool sign = v[NG - 1] > (A.M[NG - 1] / 2);
    if constexpr (sizeof(typename SH::W) == 16) {  // recombination modulo 2^128
        u128d pos = {v[0], 0};
        static_for<1, NG>([&](auto gc) { pos = add128(pos, mul_64x128(v[gc.value], A.prefix_lo[gc.value], A.prefix_hi[gc.value])); });
        const u128d full = {A.prefix_lo[NG], A.prefix_hi[NG]};
        const u128d out = sign ? sub128(pos, full) : pos;
        return typename SH::W{out.lo, out.hi};
    } else {  // words of at most 64 bits: the recombination wraps modulo 2^64
        uint64_t pos = v[0];
        static_for<1, NG>([&](auto gc) { pos += v[gc.value] * A.prefix_lo[gc.value]; });
        const uint64_t out = sign ? pos - A.prefix_lo[NG] : pos;
        return (typename SH::W)out;
    }
}

// One product (polynomial `sub` of the batch) by the TPP threads that own it.  All finished residue tiles but the last
// wait for the CRT in thread-private "parking" slots instead of registers (measured faster than carrying them at every
// size, and the only way five tiles fit at n = 4096).  Options:
//   NF_GLOBAL   the slots live in a per-workgroup region of global memory (persistent kernel below) instead of LDS:
//               n >= 8192, whose tiles no longer fit LDS next to the exchange buffer, and the ten-prime native128
//   NF_KEEP_L/R the operand is read ONCE and stays in registers for all KP primes (a re-read per prime came back from
//               HBM more often than not: rocprofv3 FETCH_SIZE showed 2.9x the operand bytes for native64 N=4096,
//               profiles/r02); otherwise it is re-read per prime (L2 / Infinity Cache hits at best)
//   NF_TW_CHUNK two twiddle loads in flight per stage instead of all (ntt_kernel.hpp stage<>): fewer live registers
//   NF_ROLL     the primes are a runtime loop instead of KP inlined copies of the three transforms
//   NF_LDS_R    (accumulating-CRT kernel) rhs is read from HBM ONCE and waits in thread-private LDS slots between the primes
enum : int { NF_GLOBAL = 1, NF_KEEP_L = 2, NF_KEEP_R = 4, NF_TW_CHUNK = 8, NF_ROLL = 16, NF_LDS_R = 32 };

template <int KIND, int LOGN, int BLK, int OPT>
__device__ __forceinline__ void native_product(typename NativeShape<KIND>::W *__restrict__ prod,
                                               const typename NativeShape<KIND>::W *__restrict__ lhs,
                                               const typename NativeShape<KIND>::W *__restrict__ rhs,
                                               const FusedTables<NativeShape<KIND>::KP> &F, const SplitArgs &S,
                                               const CrtArgs &C, uint32_t batch, uint32_t sub, uint32_t *lds_all,
                                               uint32_t *park) {
    using SH = NativeShape<KIND>;
    using W = typename SH::W;
    using Wf = NttWp<uint32_t, LOGN, false, CLS_LAZY, BLK, 1>;
    using Wi = NttWp<uint32_t, LOGN, true, CLS_LAZY, BLK, 1>;
    constexpr int E = Wf::E, TPP = Wf::TPP, NPASS = Wf::NPASS, KP = SH::KP, PARK = KP - 1;
    constexpr bool GLOBAL = (OPT & NF_GLOBAL) != 0, KEEP_L = (OPT & NF_KEEP_L) != 0, KEEP_R = (OPT & NF_KEEP_R) != 0;
    constexpr int TWC = (OPT & NF_TW_CHUNK) ? 2 : 0;
    constexpr uint32_t FULL = Wf::FULL, RM0 = Wf::S::RMASK[0];
    static_assert(RM0 == Wi::S::RMASK[NPASS - 1] && Wf::S::RMASK[NPASS - 1] == Wi::S::RMASK[0],
                  "forward and inverse schedules must mirror each other");
    static_assert(E % 4 == 0, "16-byte parking slots");
    const uint32_t tid = threadIdx.x & (TPP - 1), pl = threadIdx.x / TPP;
    uint32_t *lds = lds_all + ((size_t)pl << LOGN);
    const uint32_t subc = sub < batch ? sub : batch - 1;  // ragged tail: recompute the last polynomial, store nothing
    const W *lp = lhs + ((size_t)subc << LOGN), *rp = rhs + ((size_t)subc << LOGN);
    const uint32_t ebase = pdep<FULL & ~RM0>(tid);
    using RW = typename std::conditional<SH::BINARY, uint32_t, W>::type;
    auto load_rhs = [&](int j, uint32_t eb) -> RW {
        const W w = rp[eb | cdep((uint32_t)j, RM0)];
        if constexpr (SH::BINARY) {  // `as u32`: src/native_binary64.rs:379-385, src/native_binary128.rs:100-106
            if constexpr (sizeof(W) == 16) return (uint32_t)w.lo;
            else return (uint32_t)w;
        } else {
            return w;
        }
    };
    // global parking: tile i, coefficients 4q .. 4q+3 of this thread in one 16-byte slot, addressed as uniform base +
    // 32-bit lane offset (the saddr form of global_load / global_store).  The lane offset is opaque to the compiler: it
    // would otherwise hoist one 64-bit address per slot out of the product loop (~250 spilled registers).
    uint32_t lane16 = threadIdx.x * 16u;
    if constexpr (GLOBAL) asm volatile("" : "+v"(lane16));
    auto gpark = [&](int i, int q) -> uint4 * {
        return reinterpret_cast<uint4 *>(reinterpret_cast<char *>(park) + (size_t)(i * (E / 4) + q) * BLK * 16 + lane16);
    };
    uint32_t last[E];  // the tile of the last prime never leaves registers
    W lw[KEEP_L ? E : 1];
    RW rw[KEEP_R ? E : 1];
    if constexpr (KEEP_L) {
#pragma unroll
        for (int j = 0; j < E; ++j) lw[j] = lp[ebase | cdep((uint32_t)j, RM0)];
    }
    if constexpr (KEEP_R) {
#pragma unroll
        for (int j = 0; j < E; ++j) rw[j] = load_rhs(j, ebase);
    }
    // the inverse-transformed product residues of prime i -> a
    auto one_prime = [&](int i, uint32_t (&a)[E]) {
        uint32_t b[E];
        // The persistent kernel recomputes its twiddle and operand offsets per transform from opaque copies of the thread
        // index: shared across the inlined transforms and hoisted out of the product loop they cost hundreds of spilled
        // registers.
        uint32_t tidf = tid, tidg = tid, tidi = tid;
        if constexpr (GLOBAL) asm volatile("" : "+v"(tidf), "+v"(tidg), "+v"(tidi));
        const uint32_t eb_l = GLOBAL ? pdep<FULL & ~RM0>(tidf) : ebase, eb_r = GLOBAL ? pdep<FULL & ~RM0>(tidg) : ebase;
#pragma unroll
        for (int j = 0; j < E; ++j) a[j] = split30<W>(KEEP_L ? lw[KEEP_L ? j : 0] : lp[eb_l | cdep((uint32_t)j, RM0)], S, i);
        Wf::template pass<0, false, false, true, TWC>(a, lds, tidf, F.twf[i], nullptr, F.P[i]);
        Wf::wsync();
#pragma unroll
        for (int j = 0; j < E; ++j) {
            const RW w = KEEP_R ? rw[KEEP_R ? j : 0] : load_rhs(j, eb_r);
            if constexpr (SH::BINARY) b[j] = w;
            else b[j] = split30<W>(w, S, i);
        }
        Wf::template pass<0, false, false, true, TWC>(b, lds, tidg, F.twf[i], nullptr, F.P[i]);
#pragma unroll
        for (int j = 0; j < E; ++j) a[j] = mul_for_inv<uint32_t, CLS_LAZY>(a[j], b[j], F.P[i]);
        Wf::wsync();
        Wi::template pass<0, true, false, true, TWC>(a, lds, tidi, F.twi[i], nullptr, F.P[i]);
        Wf::wsync();
    };
    auto park_tile = [&](int i, const uint32_t (&a)[E]) {
        if constexpr (GLOBAL) {
#pragma unroll
            for (int q = 0; q < E / 4; ++q) *gpark(i, q) = uint4{a[4 * q], a[4 * q + 1], a[4 * q + 2], a[4 * q + 3]};
        } else {
#pragma unroll
            for (int j = 0; j < E; ++j) park[((size_t)i * E + j) * BLK + threadIdx.x] = a[j];
        }
    };
    // (n = 32768, 32 coefficients per thread: past hipcc's inlining budget the per-prime lambdas stay calls and everything they
    // capture -- kernel arguments, the register tiles -- moves to scratch memory; the call sites force them in.  Not for the
    // smaller sizes: forced there the LDS-parked kernels want 237 instead of 108 VGPRs.)
    constexpr bool FORCE_INLINE = E > 16;
    if constexpr ((OPT & NF_ROLL) != 0) {
        // one copy of the three transforms for the parked primes (a runtime loop: tables and constants of prime i come
        // from the kernel arguments by scalar loads) and one for the last: a fifth (a tenth for native128) of the code
#pragma clang loop unroll(disable)
        for (int i = 0; i < KP - 1; ++i) {
            uint32_t a[E];
            if constexpr (FORCE_INLINE) {
                [[clang::always_inline]] one_prime(i, a);
                [[clang::always_inline]] park_tile(i, a);
            } else {
                one_prime(i, a);
                park_tile(i, a);
            }
        }
        if constexpr (FORCE_INLINE) {
            [[clang::always_inline]] one_prime(KP - 1, last);
        } else {
            one_prime(KP - 1, last);
        }
    } else {
        static_for<0, KP>([&](auto ic) {
            constexpr int i = ic.value;
            if constexpr (i == KP - 1) {
                if constexpr (FORCE_INLINE) {
                    [[clang::always_inline]] one_prime(i, last);
                } else {
                    one_prime(i, last);
                }
            } else {
                uint32_t a[E];
                if constexpr (FORCE_INLINE) {
                    [[clang::always_inline]] one_prime(i, a);
                    [[clang::always_inline]] park_tile(i, a);
                } else {
                    one_prime(i, a);
                    park_tile(i, a);
                }
            }
        });
    }
    if (sub < batch) {
        W *op = prod + ((size_t)sub << LOGN);
        if constexpr (GLOBAL) {
            // (static_for: a rolled loop here would index `last` dynamically and send it to scratch memory)
            static_for<0, E / 4>([&](auto qc) {
                constexpr int q = qc.value;
                uint4 t[PARK > 0 ? PARK : 1];
#pragma unroll
                for (int i = 0; i < PARK; ++i) t[i] = *gpark(i, q);
                static_for<0, 4>([&](auto jc) {
                    constexpr int jj = jc.value;
                    uint32_t r[KP];
#pragma unroll
                    for (int i = 0; i < PARK; ++i) r[i] = jj == 0 ? t[i].x : jj == 1 ? t[i].y : jj == 2 ? t[i].z : t[i].w;
                    r[KP - 1] = last[4 * q + jj];
                    op[ebase | cdep((uint32_t)(4 * q + jj), RM0)] = crt_regs<SH>(r, C);
                });
                __builtin_amdgcn_sched_barrier(0);  // bounds the parked-tile loads in flight
            });
        } else {
#pragma unroll
            for (int j = 0; j < E; ++j) {
                uint32_t r[KP];
#pragma unroll
                for (int i = 0; i < PARK; ++i) r[i] = park[((size_t)i * E + j) * BLK + threadIdx.x];
                r[KP - 1] = last[j];
                op[ebase | cdep((uint32_t)j, RM0)] = crt_regs<SH>(r, C);
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// The same product with an ACCUMULATING CRT (round 4): no residue tile is parked anywhere.
//
// negacyclic_polymul's result is the wrapping image of the EXACT integer product c (|c| < n 2^(2 bits(W)), or n 2^bits(W)
// for the binary plans), and the plans' moduli M = P_0 ... P_{k-1} exceed 2 |c| by at least 2^5.9 (native64 at n = 32768:
// 2^143 against 2^149.9; every other kind / size has more room).  Any exact reconstruction of c therefore yields the
// reference's bits (src/native64.rs:91-141: its digits v0, v12, v34 and the sign test on the top digit describe the same
// integer; the edge where the top-digit test and |c| < M / 2 disagree needs |c| >= M / 2 - M / P34).  This kernel uses
//     gamma_i = r_i (M / P_i)^-1 mod P_i,     c = sum_i gamma_i (M / P_i)  -  k M,     k = round(sum_i gamma_i / P_i)
// for ANY representatives gamma_i (here in [0, 2 P_i)): per prime and coefficient
//     acc  += gamma_i * ((M / P_i) mod 2^bits(W))          (wrapping, W-wide)
//     frac += hi32(gamma_i * floor(2^59 / P_i))            (27 fractional bits: the sum is within 2^-5.9 + 2^-22 of k)
// (the pointwise product between the transforms is a lazy Montgomery product, acc_mont_lazy: its 2^-32 rides in the same constants)
// and at the end  out = acc - ((frac + 2^26) >> 27) * (M mod 2^bits(W)).  The factor (M / P_i)^-1 rides in the constants of
// the inverse transform's last stage next to 1 / n (Bfly::inv_norm: host.hip folds it into F.P[i].n_inv / last_w), so gamma_i
// IS the transform's lazy output: no canonicalisation, no multiplication.  State per coefficient: one W-wide word and one
// 32-bit word in registers instead of k - 1 parked residues; LDS holds the exchange buffer only.
// The residue split is lazy too ([0, 2 P_i): all the first butterfly stage needs): 2^32 = c_i (mod P_i) with c_i < 2^26 for
// these primes (P_i = 2^30 - d_i, c_i = 4 d_i), so a 64-bit word folds as t = hi c_i + lo < 2^58 (one v_mad_u64_u32),
// q = hi32((t >> 28) floor(2^60 / P_i)) in {floor(t / P_i) - 1, floor(t / P_i)}, r = lo32(t) - q P_i.
// Public fwd() / inv() of the plans keep split_kernel / crt_kernel: they take and produce arbitrary canonical residues.
// ---------------------------------------------------------------------------------------------------------------
struct AccArgs {
    uint64_t c_lo[10], c_hi[10];  // (M / P_i) mod 2^128
    uint64_t m_lo, m_hi;          // M mod 2^128
    uint32_t f[10];               // floor(2^59 / P_i)
    uint32_t m60[10];             // floor(2^60 / P_i)
};
constexpr int ACC_FRAC_BITS = 27;
constexpr int ACC_FAM = 3;   // schedule family of the transforms inside native_product_acc (gen_sched.py)

// t < 2^58 -> t mod p in [0, 2p)
__device__ __forceinline__ uint32_t acc_red58(uint64_t t, uint32_t p, uint32_t m60) {
    const uint32_t q = __umulhi((uint32_t)(t >> 28), m60);
    return (uint32_t)mad_box<true>(q, 0u - p, t);   // lo32(t) - q p as ONE v_mad_u64_u32 (asm: hipcc narrows the C++ form to mul_lo + sub)
}
// a * b / 2^32 mod p in [0, 2p) for lazy a, b in [0, 4p), p < 2^30 (Montgomery; pinv_neg = -p^-1 mod 2^32): each operand takes one
// conditional subtraction of 2p (a b < 4 p^2 < 2^62, t + m p < 2^63, u < p^2 / 2^30 + p < 2p) -- seven instructions where the
// canonical route (two canonicalisations + Barrett, mul_for_inv) takes thirteen.  The factor 2^-32 is undone by the constants of the
// inverse transform's last stage (host.hip, build_acc_args).
__device__ __forceinline__ uint32_t acc_mont_lazy(uint32_t a, uint32_t b, const ModParams<uint32_t> &P) {
    a = umin<uint32_t>(a, a - P.two_p);
    b = umin<uint32_t>(b, b - P.two_p);
    const uint64_t t = (uint64_t)a * b;
    const uint32_t m = (uint32_t)t * P.pinv_neg;
    return (uint32_t)(((uint64_t)m * P.p + t) >> 32);
}
// value mod P_k in [0, 2 P_k) for a u32 / u64 / u128 word
// KEPT: the word stays in registers across the primes (NF_KEEP_L)
template <class W, bool KEPT = false> __device__ __forceinline__ uint32_t split30_lazy(W w, const SplitArgs &A, const AccArgs &C, int k) {
    const uint32_t p = (uint32_t)A.prime[k], c = A.c[k], m60 = C.m60[k];
    if constexpr (sizeof(W) == 4) {
        return red32_lazy((uint32_t)w, p, A.one_shoup[k]);
    } else if constexpr (sizeof(W) == 8 && !KEPT) {
        return acc_red58((uint64_t)(uint32_t)((uint64_t)w >> 32) * c + (uint32_t)w, p, m60);
    } else if constexpr (sizeof(W) == 8) {
        // a word that is re-used by every prime: t = hi c + lo with the WORD ITSELF as the multiply-add's addend: hi c + (lo + hi 2^32), then hi taken off the upper word again.
        // (Written as hi * c + zext(lo), hipcc builds a (lo, 0) register pair per kept lhs word outside the prime loop: 48 registers for
        // the 16 words instead of 32 -- the n = 2048 shapes spilled on it, round 5.  Words that are re-read per prime keep the plain form: the
        // 128-VGPR shapes of n = 8192 / 16384 spill 9 registers with this one.)
        const uint32_t hi = (uint32_t)((uint64_t)w >> 32);
        uint64_t t;
        asm("v_mad_u64_u32 %0, vcc, %1, %2, %3" : "=v"(t) : "v"(hi), "s"(c), "v"((uint64_t)w) : "vcc");
        const uint32_t th = (uint32_t)(t >> 32) - hi, tl = (uint32_t)t;
        const uint32_t q = __umulhi(__builtin_amdgcn_alignbit(th, tl, 28), m60);
        return (uint32_t)mad_box<true>(q, 0u - p, t);   // low word: lo32(t) - q p (the upper word of the addend does not reach it)
    } else {
        uint32_t r = acc_red58((uint64_t)(uint32_t)(w.hi >> 32) * c + (uint32_t)w.hi, p, m60);   // r < 2^31: r c + limb < 2^58
        r = acc_red58((uint64_t)r * c + (uint32_t)(w.lo >> 32), p, m60);
        return acc_red58((uint64_t)r * c + (uint32_t)w.lo, p, m60);
    }
}

template <class W> struct AccWord;
template <> struct AccWord<uint32_t> {
    using A = uint32_t;
    static __device__ __forceinline__ A mad(A acc, uint32_t g, uint64_t lo, uint64_t) { return acc + g * (uint32_t)lo; }
    static __device__ __forceinline__ void pin(A &acc) { asm volatile("" : "+v"(acc)); }
    static __device__ __forceinline__ uint32_t out(A acc, uint32_t k, uint64_t mlo, uint64_t) { return acc - k * (uint32_t)mlo; }
};
template <> struct AccWord<uint64_t> {
    using A = uint64_t;
    // acc + g * lo (mod 2^64): one v_mad_u64_u32 for the low word of the constant, v_mul_lo_u32 + v_add_u32 for the high one (asm: left to
    // itself hipcc turns the latter into a second v_mad_u64_u32 on a register pair it has to build and take apart with two moves)
    static __device__ __forceinline__ A mad(A acc, uint32_t g, uint64_t lo, uint64_t) {
        const uint64_t t = (uint64_t)g * (uint32_t)lo + acc;
        uint32_t h;
        asm("v_mul_lo_u32 %0, %1, %2" : "=v"(h) : "v"(g), "s"((uint32_t)(lo >> 32)));
        return ((uint64_t)((uint32_t)(t >> 32) + h) << 32) | (uint32_t)t;
    }
    static __device__ __forceinline__ void pin(A &acc) { asm volatile("" : "+v"(acc)); }
    static __device__ __forceinline__ uint64_t out(A acc, uint32_t k, uint64_t mlo, uint64_t) { return acc - (uint64_t)k * mlo; }
};
template <> struct AccWord<Word128> {
    using A = unsigned __int128;
    static __device__ __forceinline__ A mad(A acc, uint32_t g, uint64_t lo, uint64_t hi) {
        return acc + (A)g * (((A)hi << 64) | lo);
    }
    static __device__ __forceinline__ void pin(A &acc) {
        uint64_t l = (uint64_t)acc, h = (uint64_t)(acc >> 64);
        asm volatile("" : "+v"(l), "+v"(h));
        acc = ((A)h << 64) | l;
    }
    static __device__ __forceinline__ Word128 out(A acc, uint32_t k, uint64_t mlo, uint64_t mhi) {
        const A r = acc - (A)k * (((A)mhi << 64) | mlo);
        return Word128{(uint64_t)r, (uint64_t)(r >> 64)};
    }
};

// NF_KEEP_L / NF_KEEP_R / NF_TW_CHUNK as native_product
template <int KIND, int LOGN, int BLK, int OPT>
__device__ __forceinline__ void native_product_acc(typename NativeShape<KIND>::W *__restrict__ prod,
                                                   const typename NativeShape<KIND>::W *__restrict__ lhs,
                                                   const typename NativeShape<KIND>::W *__restrict__ rhs,
                                                   const FusedTables<NativeShape<KIND>::KP> &F, const SplitArgs &S,
                                                   const AccArgs &C, uint32_t batch, uint32_t sub0, uint32_t *lds_all,
                                                   void *rstash_all = nullptr) {
    using SH = NativeShape<KIND>;
    using W = typename SH::W;
    using AW = AccWord<W>;
    using Wf = NttWp<uint32_t, LOGN, false, CLS_LAZY, BLK, ACC_FAM>;   // padded exchange layout: no address arithmetic per access
    using Wi = NttWp<uint32_t, LOGN, true, CLS_LAZY, BLK, ACC_FAM>;
    constexpr int E = Wf::E, TPP = Wf::TPP, NPASS = Wf::NPASS, KP = SH::KP;
    constexpr bool KEEP_L = (OPT & NF_KEEP_L) != 0, KEEP_R = (OPT & NF_KEEP_R) != 0;
    constexpr int TWC = (OPT & NF_TW_CHUNK) ? 2 : 0;
    constexpr uint32_t FULL = Wf::FULL, RM0 = Wf::S::RMASK[0];
    static_assert(RM0 == Wi::S::RMASK[NPASS - 1] && Wf::S::RMASK[NPASS - 1] == Wi::S::RMASK[0],
                  "forward and inverse schedules must mirror each other");
    static_assert(2ull * KP * (1ull << ACC_FRAC_BITS) + (1ull << (ACC_FRAC_BITS - 1)) <= (1ull << 32),
                  "the fraction sum of lazy residues (each below 2 P_i) and its rounding constant fit 32 bits");
    const uint32_t tid = threadIdx.x & (TPP - 1), pl = threadIdx.x / TPP;
    uint32_t *lds = lds_all + (size_t)pl * Wf::B::LDS_WORDS_1;
    const uint32_t sub = sub0 + pl;
    // operands: workgroup-uniform base + one 32-bit byte offset per thread + the instruction's immediate (ragged tail: the threads of
    // products past the end recompute the last one and store nothing)
    const uint32_t plc = sub < batch ? pl : batch - 1 - sub0;
    const W *lp0 = lhs + ((size_t)sub0 << LOGN), *rp0 = rhs + ((size_t)sub0 << LOGN);
    auto word_at = [&](const W *base0, int j, uint32_t eb) -> W {
        const uint32_t cb = cdep((uint32_t)j, RM0) * (uint32_t)sizeof(W), win = cb & ~4095u, imm = cb & 4095u;
        const char *base = reinterpret_cast<const char *>(base0) + win;
        return *reinterpret_cast<const W *>(base + (size_t)(((plc << LOGN) + eb) * (uint32_t)sizeof(W)) + imm);
    };
    using RW = typename std::conditional<SH::BINARY, uint32_t, W>::type;
    auto load_rhs = [&](int j, uint32_t eb) -> RW {
        const W w = word_at(rp0, j, eb);
        if constexpr (SH::BINARY) {  // `as u32`: src/native_binary64.rs:379-385, src/native_binary128.rs:100-106
            if constexpr (sizeof(W) == 16) return (uint32_t)w.lo;
            else return (uint32_t)w;
        } else {
            return w;
        }
    };
    W lw[KEEP_L ? E : 1];
    RW rw[KEEP_R ? E : 1];
    // NF_LDS_R: the rhs words of this thread, read from HBM once, wait in LDS slots only this thread touches ([j][thread]: conflict-free,
    // no synchronisation) -- round 4 re-read rhs from HBM / L2 once per prime (PMC: 1.50 x the algorithmic bytes for C3, 1.34 x for C5)
    constexpr bool LDS_R = (OPT & NF_LDS_R) != 0 && !KEEP_R;
    RW *rst = reinterpret_cast<RW *>(rstash_all) + threadIdx.x;
    {
        const uint32_t ebase = pdep<FULL & ~RM0>(tid);
        if constexpr (LDS_R) {
            // first, and fenced off from the lhs loads: in flight together the two operands' words do not fit the register budget (the
            // n = 2048 shapes spilled 2 ... 4 registers once per thread -- 7 % extra HBM traffic on C5 through scratch memory)
#pragma unroll
            for (int j = 0; j < E; ++j) rst[j * BLK] = load_rhs(j, ebase);
            __builtin_amdgcn_sched_barrier(0);
        }
        if constexpr (KEEP_L) {
#pragma unroll
            for (int j = 0; j < E; ++j) lw[j] = word_at(lp0, j, ebase);
        }
        if constexpr (KEEP_R) {
#pragma unroll
            for (int j = 0; j < E; ++j) rw[j] = load_rhs(j, ebase);
        }
    }
    typename AW::A acc[E];
    uint32_t frac[E];
#pragma unroll
    for (int j = 0; j < E; ++j) {
        acc[j] = 0;
        frac[j] = 0;
    }
    auto one_prime = [&](const int i) {
        uint32_t a[E], b[E];
        // offsets of the three transforms recomputed per prime from opaque copies of the thread index: hoisted out of the
        // prime loop (k x 3 inlined transforms) they would stay live next to the accumulators
        uint32_t tidf = tid, tidg = tid, tidi = tid;
        asm volatile("" : "+v"(tidf), "+v"(tidg), "+v"(tidi));
        const uint32_t eb_l = pdep<FULL & ~RM0>(tidf), eb_r = pdep<FULL & ~RM0>(tidg);
        // (2p and p copied into VECTOR registers -- plain v_add_u32 / v_sub_u32 on vector operands issue at twice the rate of the
        // scalar-operand forms in isolation, profiles/r04_ubench3_valu_forms.txt -- made the kernel 5 % SLOWER: tools/native_lab)
        const ModParams<uint32_t> &Pv = F.P[i];
#pragma unroll
        for (int j = 0; j < E; ++j)
            a[j] = split30_lazy<W, KEEP_L>(KEEP_L ? lw[KEEP_L ? j : 0] : word_at(lp0, j, eb_l), S, C, i);
        __builtin_amdgcn_sched_barrier(0);   // (the phases of a prime, and the primes, kept apart: overlapped they spill)
        Wf::template pass<0, false, false, false, TWC>(a, lds, tidf, F.twf[i], nullptr, Pv);   // FIN = false: lazy outputs in [0, 4p)
        Wf::wsync();
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int j = 0; j < E; ++j) {
            RW w;
            if constexpr (KEEP_R) w = rw[KEEP_R ? j : 0];
            else if constexpr (LDS_R) w = rst[j * BLK];
            else w = load_rhs(j, eb_r);
            if constexpr (SH::BINARY) b[j] = w;
            else b[j] = split30_lazy<W>(w, S, C, i);
        }
        __builtin_amdgcn_sched_barrier(0);
        Wf::template pass<0, false, false, false, TWC>(b, lds, tidg, F.twf[i], nullptr, Pv);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int j = 0; j < E; ++j) a[j] = acc_mont_lazy(a[j], b[j], Pv);
        Wf::wsync();
        __builtin_amdgcn_sched_barrier(0);
        // FIN = false: the lazy outputs in [0, 2 P_i) are gamma_i (the last stage's constants carry (M / P_i)^-1 / n)
        Wi::template pass<0, true, false, false, TWC>(a, lds, tidi, F.twi[i], nullptr, Pv);
        __builtin_amdgcn_sched_barrier(0);
        const uint64_t clo = C.c_lo[i], chi = C.c_hi[i];
        const uint32_t fi = C.f[i];
#pragma unroll
        for (int j = 0; j < E; ++j) {
            acc[j] = AW::mad(acc[j], a[j], clo, chi);
            frac[j] += __umulhi(a[j], fi);
            // (opaque: left alone hipcc sinks the whole sum to the final store and keeps -- spills -- the k residue tiles instead
            // of the accumulators, i.e. re-invents the parking this kernel exists to avoid)
            AW::pin(acc[j]);
            asm volatile("" : "+v"(frac[j]));
        }
        Wf::wsync();  // the exchange buffer is reused by the next prime
        __builtin_amdgcn_sched_barrier(0);
    };
    if constexpr ((OPT & NF_ROLL) != 0) {
        // the primes as a RUNTIME loop: one copy of the three transforms (tables and constants of prime i by scalar loads from the
        // kernel arguments) -- a k-th of the code
#pragma clang loop unroll(disable)
        for (int i = 0; i < KP; ++i) one_prime(i);
    } else {
        static_for<0, KP>([&](auto ic) { one_prime(ic.value); });
    }
    // the output addresses from an opaque copy of the thread index: computed at the top of the kernel (where hipcc would put them)
    // they are sixteen 64-bit values that live -- spilled -- through all k primes
    uint32_t tx = threadIdx.x;
    asm volatile("" : "+v"(tx));
    const uint32_t subo = sub0 + tx / TPP;
    if (subo < batch) {
        W *op = prod + ((size_t)subo << LOGN);
        const uint32_t ebo = pdep<FULL & ~RM0>(tx & (TPP - 1));
#pragma unroll
        for (int j = 0; j < E; ++j) {
            const uint32_t k = (frac[j] + (1u << (ACC_FRAC_BITS - 1))) >> ACC_FRAC_BITS;
            op[ebo | cdep((uint32_t)j, RM0)] = AW::out(acc[j], k, C.m_lo, C.m_hi);
        }
    }
}

template <int KIND, int LOGN, int BLK, int WPS, int OPT>
__global__ __launch_bounds__(BLK, WPS) void native_polymul_kernel_acc(typename NativeShape<KIND>::W *__restrict__ prod,
                                                                  const typename NativeShape<KIND>::W *__restrict__ lhs,
                                                                  const typename NativeShape<KIND>::W *__restrict__ rhs,
                                                                  const FusedTables<NativeShape<KIND>::KP> F,
                                                                  const SplitArgs S, const AccArgs C, uint32_t batch) {
    using Wf = NttWp<uint32_t, LOGN, false, CLS_LAZY, BLK, ACC_FAM>;
    constexpr int PPB = BLK / Wf::TPP;
    static_assert(BLK % Wf::TPP == 0 && PPB >= 1, "whole products per workgroup");
    __shared__ __attribute__((aligned(16))) uint32_t lds_all[(size_t)PPB * Wf::B::LDS_WORDS_1];
    using SH = NativeShape<KIND>;
    using RW = typename std::conditional<SH::BINARY, uint32_t, typename SH::W>::type;
    constexpr bool LDS_R = (OPT & NF_LDS_R) != 0 && (OPT & NF_KEEP_R) == 0;
    __shared__ __attribute__((aligned(16))) RW rstash[LDS_R ? (size_t)Wf::E * BLK : 1];
    native_product_acc<KIND, LOGN, BLK, OPT>(prod, lhs, rhs, F, S, C, batch, blockIdx.x * PPB, lds_all, rstash);
}

// products per workgroup / parked words per workgroup of a shape
template <int KIND, int LOGN, int BLK> struct NativeTile {
    using Wf = NttWp<uint32_t, LOGN, false, CLS_LAZY, BLK, 1>;
    static constexpr int E = Wf::E, PPB = BLK / Wf::TPP, PARK = NativeShape<KIND>::KP - 1;
    static constexpr size_t PARK_WORDS = (size_t)PARK * BLK * E;
    static_assert(BLK % Wf::TPP == 0 && PPB >= 1, "whole products per workgroup");
};

// n <= 4096: one workgroup per PPB products, parked tiles in LDS, both operands kept in registers
template <int KIND, int LOGN, int BLK, int OPT = 0>
__global__ __launch_bounds__(BLK, 2) void native_polymul_kernel(typename NativeShape<KIND>::W *__restrict__ prod,
                                                             const typename NativeShape<KIND>::W *__restrict__ lhs,
                                                             const typename NativeShape<KIND>::W *__restrict__ rhs,
                                                             const FusedTables<NativeShape<KIND>::KP> F, const SplitArgs S,
                                                             const CrtArgs C, uint32_t batch) {
    using T = NativeTile<KIND, LOGN, BLK>;
    __shared__ __attribute__((aligned(16))) uint32_t lds_all[(size_t)T::PPB << LOGN];
    __shared__ uint32_t park[T::PARK > 0 ? T::PARK_WORDS : 1];
    native_product<KIND, LOGN, BLK, NF_KEEP_L | NF_KEEP_R | OPT>(prod, lhs, rhs, F, S, C, batch,
                                                           blockIdx.x * T::PPB + threadIdx.x / T::Wf::TPP, lds_all, park);
}

// Persistent form: a grid sized to the device, every workgroup working through PPB products at a time and parking their
// residue tiles in its own region of a global scratch area (grid x PARK_WORDS words, reused product after product, so it
// stays in L2 / Infinity Cache).  n = 8192 / 16384 of every fused kind, and native128 (ten primes) at every size.
template <int KIND, int LOGN, int BLK, int WPS, int OPT>
__global__ __launch_bounds__(BLK, WPS) void native_polymul_kernel_g(typename NativeShape<KIND>::W *__restrict__ prod,
                                                                const typename NativeShape<KIND>::W *__restrict__ lhs,
                                                                const typename NativeShape<KIND>::W *__restrict__ rhs,
                                                                const FusedTables<NativeShape<KIND>::KP> F,
                                                                const SplitArgs S, const CrtArgs C, uint32_t batch,
                                                                uint32_t *__restrict__ scratch) {
    using T = NativeTile<KIND, LOGN, BLK>;
    __shared__ __attribute__((aligned(16))) uint32_t lds_all[(size_t)T::PPB << LOGN];
    uint32_t *park = scratch + (size_t)blockIdx.x * T::PARK_WORDS;
    const uint32_t groups = (batch + T::PPB - 1) / T::PPB;
    for (uint32_t g = blockIdx.x; g < groups; g += gridDim.x) {
        native_product<KIND, LOGN, BLK, OPT | NF_GLOBAL>(prod, lhs, rhs, F, S, C, batch, g * T::PPB + threadIdx.x / T::Wf::TPP,
                                                         lds_all, park);
        T::Wf::wsync();  // the exchange buffer and the parked slots are reused by the next product
    }
}

}  // namespace cntt
