// LDS-resident batched negacyclic NTT kernel for gfx950 (one code path for u32/u64, fwd/inv).
//
// Replaces the reference's transform engines
//   fwd: src/prime64/shoup.rs:10-706, src/prime32/shoup.rs:9-708, src/prime64/generic_solinas.rs:449-481
//   inv: src/prime64/shoup.rs:710-1468, src/prime32/shoup.rs:712-1481, generic_solinas.rs:483-561
// Same schedule as the reference (stage on index bit b uses table entry 2^(LOGN-1-b) + (e >> (b+1)),
// forward visits b descending, inverse ascending; no 1/N in the inverse), evaluated as a few
// register-resident radix-2^LOGE passes with LDS exchanges in between (see gen_sched.py).
//
// Work decomposition: a polynomial is owned by TPP = N / 2^LOGE consecutive threads, so for
// N <= 64 * 2^LOGE it lives inside ONE 64-lane wavefront and the exchanges need no workgroup
// barrier (LDS operations of one wave execute in order); larger N use __syncthreads().
#pragma once
#include <type_traits>

#include "ntt_arith.hpp"

namespace cntt {

// FAM 0: the tuned schedule.  FAM 1: 16 coefficients per thread wherever FAM 0 uses 32 (u32 N=2048/4096) -- for
// kernels that keep several residue tiles in registers; identical to FAM 0 for every other size.
template <int BITS, int LOGN, bool INV, int FAM = 0> struct Sched;  // specialisations: sched_gen.inc
#include "sched_gen.inc"
template <int BITS, int LOGN, bool INV> struct Sched<BITS, LOGN, INV, 1> : Sched<BITS, LOGN, INV, 0> {};

// ---- compile-time bit helpers ---------------------------------------------------------------
__host__ __device__ constexpr int cpop(uint32_t m) {
    int c = 0;
    for (; m; m &= m - 1) ++c;
    return c;
}
// rank of bit b inside mask m (number of set bits of m below b)
__host__ __device__ constexpr int crank(uint32_t m, int b) { return cpop(m & ((1u << b) - 1u)); }
// compile-time pdep for constants
__host__ __device__ constexpr uint32_t cdep(uint32_t x, uint32_t mask) {
    uint32_t out = 0;
    int k = 0;
    for (int b = 0; b < 32; ++b)
        if ((mask >> b) & 1u) {
            out |= ((x >> k) & 1u) << b;
            ++k;
        }
    return out;
}
// number of low contiguous set bits (bits 0..v-1 all set)
__host__ __device__ constexpr int clow(uint32_t m) {
    int v = 0;
    while ((m >> v) & 1u) ++v;
    return v;
}
// runtime pdep with a compile-time mask: one shift/and/or per run of consecutive mask bits
__host__ __device__ constexpr int crun(uint32_t m, int b) {  // length of the run of set bits starting at b
    int len = 0;
    while (b + len < 32 && ((m >> (b + len)) & 1u)) ++len;
    return len;
}
template <uint32_t MASK, int B = 0, int K = 0> __device__ __forceinline__ uint32_t pdep(uint32_t x) {
    if constexpr (B >= 32) {
        return 0u;
    } else if constexpr ((MASK >> B) == 0u) {
        return 0u;
    } else if constexpr (((MASK >> B) & 1u) == 0u) {
        return pdep<MASK, B + 1, K>(x);
    } else {
        constexpr int LEN = crun(MASK, B);
        constexpr uint32_t LM = (LEN >= 32) ? 0xffffffffu : ((1u << LEN) - 1u);
        return (((x >> K) & LM) << B) | pdep<MASK, B + LEN, K + LEN>(x);
    }
}

template <class T, int NV> struct VecOf;
template <> struct VecOf<uint64_t, 1> { using type = uint64_t; };
template <> struct VecOf<uint64_t, 2> { using type = __attribute__((ext_vector_type(2))) uint64_t; };
template <> struct VecOf<uint32_t, 1> { using type = uint32_t; };
template <> struct VecOf<uint32_t, 2> { using type = __attribute__((ext_vector_type(2))) uint32_t; };
template <> struct VecOf<uint32_t, 4> { using type = __attribute__((ext_vector_type(4))) uint32_t; };

// ---- the kernel -----------------------------------------------------------------------------
template <class T, int LOGN, bool INV, int CLS, bool SUB, int FAM = 0>
struct NttKernel {
    using elem_t = T;
    static constexpr int BITS = sizeof(T) * 8;
    using S = Sched<BITS, LOGN, INV, FAM>;
    static constexpr int LOGE = S::LOGE, E = 1 << LOGE, NPASS = S::NPASS, BLOCK = S::BLOCK;
    static constexpr int TPP = 1 << (LOGN - LOGE);
    static constexpr int PPB = (BLOCK / TPP) > 0 ? (BLOCK / TPP) : 1;
    static constexpr uint32_t FULL = (LOGN >= 32) ? 0xffffffffu : ((1u << LOGN) - 1u);
    static constexpr bool WAVE_PRIVATE = TPP <= 64;  // a polynomial never spans two wavefronts
    static constexpr int MAXV = 16 / (int)sizeof(T);  // elements per 16-byte access

    // LDS layout of the exchange buffer: an XOR swizzle of the element index, or (schedule family 2) a PADDED layout
    // e + (e >> PAD_SH) * PAD_MUL.  The padded one is additive over disjoint index bits, so gather / scatter add the
    // register part as a compile-time constant (the DS instruction's immediate offset) to one per-thread base.
    static constexpr bool PADDED = S::PAD_MUL != 0;
    static_assert(!PADDED || (S::SWZ_M0 == 0 && S::SWZ_M1 == 0), "a layout is padded or swizzled");
    static __host__ __device__ constexpr uint32_t pad_of(uint32_t e) { return e + (e >> S::PAD_SH) * S::PAD_MUL; }
    static constexpr size_t LDS_WORDS_1 = PADDED ? (size_t)pad_of(1u << LOGN) : ((size_t)1 << LOGN);   // one polynomial
    static constexpr size_t LDS_ELEMS = (NPASS > 1) ? (size_t)PPB * LDS_WORDS_1 : 1;
    static __device__ __forceinline__ uint32_t phys(uint32_t e) {
        if constexpr (PADDED) return pad_of(e);
        uint32_t o = e;
        if constexpr (S::SWZ_M0 != 0) o ^= ((e >> S::SWZ_SH0) & S::SWZ_M0) << S::SWZ_L0;
        if constexpr (S::SWZ_M1 != 0) o ^= ((e >> S::SWZ_SH1) & S::SWZ_M1) << S::SWZ_L1;
        return o;
    }

    static __device__ __forceinline__ void sync() {
        if constexpr (WAVE_PRIVATE) {
            // Same-wave LDS operations execute in issue order, so only the COMPILER must not reorder
            // them.  A fence builtin would also make hipcc drain vmcnt(0) here, which would serialise the
            // prefetched global loads of the persistent kernel behind every exchange.
            asm volatile("" ::: "memory");
            __builtin_amdgcn_wave_barrier();
            asm volatile("" ::: "memory");
        } else {
            __syncthreads();
        }
    }

    // vector width (elements) usable for a layout: low contiguous register bits, at most 16 bytes
    template <uint32_t RM> static constexpr int vec_elems() {
        int v = clow(RM);
        int n = 1 << v;
        return n > MAXV ? MAXV : n;
    }

    template <uint32_t RM, class PTR> static __device__ __forceinline__ void gather(T (&r)[E], PTR base, uint32_t ebase, bool lds) {
        constexpr int NV = vec_elems<RM>();
        using V = typename VecOf<T, NV>::type;
        const uint32_t pbase = (PADDED && lds) ? pad_of(ebase) : 0u;
#pragma unroll
        for (int j = 0; j < E; j += NV) {
            const uint32_t e = ebase | cdep((uint32_t)j, RM);
            const uint32_t a = lds ? (PADDED ? pbase + pad_of(cdep((uint32_t)j, RM)) : phys(e)) : e;
            if constexpr (NV == 1) {
                r[j] = base[a];
            } else {
                const V v = *reinterpret_cast<const V *>(&base[a]);
#pragma unroll
                for (int i = 0; i < NV; ++i) r[j + i] = v[i];
            }
        }
    }
    template <uint32_t RM, class PTR> static __device__ __forceinline__ void scatter(const T (&r)[E], PTR base, uint32_t ebase, bool lds) {
        constexpr int NV = vec_elems<RM>();
        using V = typename VecOf<T, NV>::type;
        const uint32_t pbase = (PADDED && lds) ? pad_of(ebase) : 0u;
#pragma unroll
        for (int j = 0; j < E; j += NV) {
            const uint32_t e = ebase | cdep((uint32_t)j, RM);
            const uint32_t a = lds ? (PADDED ? pbase + pad_of(cdep((uint32_t)j, RM)) : phys(e)) : e;
            if constexpr (NV == 1) {
                base[a] = r[j];
            } else {
                V v;
#pragma unroll
                for (int i = 0; i < NV; ++i) v[i] = r[j + i];
                *reinterpret_cast<V *>(&base[a]) = v;
            }
        }
    }

    // ---- tile-relative global addressing (persistent kernels) -----------------------------------
    // NT (round 5): the non-temporal hint for batches that stream through the chip once (larger than the 256 MiB Infinity Cache:
    // ModParams::stream, decided per launch by the host).  On the asynchronous tile loads AND the tile stores of the stand-alone
    // transforms together it is worth -3 % (either alone nothing; nothing in the fused product, whose rhs loads are synchronous and get
    // slower with it: profiles/r05_nt_hint_ab.txt, r05_nt_headline_ab.txt); on a batch that fits the cache it costs 3 ... 6 %, because the
    // next kernel then finds nothing there (profiles/r05_small_batch_ab.txt).  The plan tables always keep the default policy.
    // A tile's address is a workgroup-uniform base (an SGPR pair) plus ONE 32-bit byte offset per thread: the
    // compiler selects global_load / global_store with an saddr operand, and no 64-bit per-thread pointers live in
    // VGPRs across the butterflies.
    template <uint32_t RM> static __device__ __forceinline__ void gather_tile(T (&r)[E], const T *tile, uint32_t voff) {
        constexpr int NV = vec_elems<RM>();
        using V = typename VecOf<T, NV>::type;
#pragma unroll
        for (int j = 0; j < E; j += NV) {
            const uint32_t off = voff + cdep((uint32_t)j, RM) * (uint32_t)sizeof(T);
            if constexpr (NV == 1) {
                r[j] = *reinterpret_cast<const T *>(reinterpret_cast<const char *>(tile) + off);
            } else {
                const V v = *reinterpret_cast<const V *>(reinterpret_cast<const char *>(tile) + off);
#pragma unroll
                for (int i = 0; i < NV; ++i) r[j + i] = v[i];
            }
        }
    }
    // elements [J0, J0 + CNT) of the same layout (fused kernels that take an operand a few vectors at a time)
    template <uint32_t RM, int J0, int CNT> static __device__ __forceinline__ void gather_tile_part(T (&r)[CNT], const T *tile, uint32_t voff) {
        constexpr int NV = vec_elems<RM>();
        using V = typename VecOf<T, NV>::type;
        static_assert(J0 % NV == 0 && CNT % NV == 0, "whole vectors");
#pragma unroll
        for (int j = 0; j < CNT; j += NV) {
            const uint32_t off = voff + cdep((uint32_t)(J0 + j), RM) * (uint32_t)sizeof(T);
            if constexpr (NV == 1) {
                r[j] = *reinterpret_cast<const T *>(reinterpret_cast<const char *>(tile) + off);
            } else {
                const V v = *reinterpret_cast<const V *>(reinterpret_cast<const char *>(tile) + off);
#pragma unroll
                for (int i = 0; i < NV; ++i) r[j + i] = v[i];
            }
        }
    }
    template <uint32_t RM, bool NT = false> static __device__ __forceinline__ void scatter_tile(const T (&r)[E], T *tile, uint32_t voff) {
        constexpr int NV = vec_elems<RM>();
        using V = typename VecOf<T, NV>::type;
#pragma unroll
        for (int j = 0; j < E; j += NV) {
            const uint32_t off = voff + cdep((uint32_t)j, RM) * (uint32_t)sizeof(T);
            if constexpr (NV == 1) {
                if constexpr (NT) __builtin_nontemporal_store(r[j], reinterpret_cast<T *>(reinterpret_cast<char *>(tile) + off));
                else *reinterpret_cast<T *>(reinterpret_cast<char *>(tile) + off) = r[j];
            } else {
                V v;
#pragma unroll
                for (int i = 0; i < NV; ++i) v[i] = r[j + i];
                if constexpr (NT) __builtin_nontemporal_store(v, reinterpret_cast<V *>(reinterpret_cast<char *>(tile) + off));
                else *reinterpret_cast<V *>(reinterpret_cast<char *>(tile) + off) = v;
            }
        }
    }
    // the same with the policy chosen at run time (a workgroup-uniform flag: one scalar branch around the tile's stores)
    template <uint32_t RM> static __device__ __forceinline__ void scatter_tile_rt(const T (&r)[E], T *tile, uint32_t voff, bool nt) {
        if (nt) scatter_tile<RM, true>(r, tile, voff);
        else scatter_tile<RM, false>(r, tile, voff);
    }

    // ---- software-pipelined global loads (persistent kernel) ------------------------------------
    // hipcc keeps ONE in-order vmcnt model per loop and waits for freshly issued loads at the loop header,
    // so a prefetch written in plain C++ is waited for at once.  These loads are issued from inline asm
    // (invisible to the compiler's counters); wait_async() is the single hand-placed counted wait.  The
    // destination vectors must not be read between the two (they are only passed to wait_async), and a kernel
    // that uses them must not spill: a destination register spilled or reassigned while the load is in flight
    // is overwritten when the data lands (tests/test_host_plan.py checks the code objects for zero spills).
    static constexpr int ASYNC_NV = MAXV;  // elements per 16-byte load
    using AsyncVec = __attribute__((ext_vector_type(4))) uint32_t;
    template <uint32_t RM> static constexpr bool async_ok() { return vec_elems<RM>() == MAXV; }

    // The loads use the saddr form (SGPR-pair base + 32-bit VGPR byte offset + 13-bit immediate): the per-vector
    // element offsets cdep(j, RM) are compile-time constants.
    template <uint32_t RM, int JV = 0, bool NT = false>
    static __device__ __forceinline__ void gather_async(AsyncVec (&v)[E / MAXV], const T *tile, uint32_t voff) {
        if constexpr (JV < E / MAXV) {
            // the 4 KiB window of a vector goes into the scalar base (SALU add), the rest into the immediate: one VGPR
            // offset serves the whole tile
            constexpr uint32_t BYTE = cdep((uint32_t)(JV * MAXV), RM) * (uint32_t)sizeof(T);
            constexpr uint32_t WIN = BYTE & ~4095u, IMM = BYTE & 4095u;
            const char *base = reinterpret_cast<const char *>(tile) + WIN;
            if constexpr (NT) asm volatile("global_load_dwordx4 %0, %1, %2 offset:%3 nt" : "=v"(v[JV]) : "v"(voff), "s"(base), "n"(IMM) : "memory");
            else asm volatile("global_load_dwordx4 %0, %1, %2 offset:%3" : "=v"(v[JV]) : "v"(voff), "s"(base), "n"(IMM) : "memory");
            gather_async<RM, JV + 1, NT>(v, tile, voff);
        }
    }
    template <uint32_t RM>
    static __device__ __forceinline__ void gather_async_rt(AsyncVec (&v)[E / MAXV], const T *tile, uint32_t voff, bool nt) {
        if (nt) gather_async<RM, 0, true>(v, tile, voff);
        else gather_async<RM, 0, false>(v, tile, voff);
    }
    // wait until at most YOUNGER vector-memory operations (issued after the async loads) are outstanding
    template <int YOUNGER> static __device__ __forceinline__ void wait_async(AsyncVec (&v)[E / MAXV]) {
        static_assert(E / MAXV == 8 || E / MAXV == 4 || E / MAXV == 16, "unexpected register tile");
        if constexpr (E / MAXV == 8) {
            asm volatile("s_waitcnt vmcnt(%8)"
                         : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7])
                         : "n"(YOUNGER)
                         : "memory");
        } else if constexpr (E / MAXV == 4) {
            asm volatile("s_waitcnt vmcnt(%4)" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]) : "n"(YOUNGER) : "memory");
        } else {  // every destination vector is a tied operand of the ONE waiting statement
            asm volatile("s_waitcnt vmcnt(%16)"
                         : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7]),
                           "+v"(v[8]), "+v"(v[9]), "+v"(v[10]), "+v"(v[11]), "+v"(v[12]), "+v"(v[13]), "+v"(v[14]), "+v"(v[15])
                         : "n"(YOUNGER)
                         : "memory");
        }
    }
    static __device__ __forceinline__ void unpack_async(T (&r)[E], const AsyncVec (&v)[E / MAXV]) {
#pragma unroll
        for (int j = 0; j < E; j += MAXV) {
            if constexpr (sizeof(T) == 8) {
                r[j] = (T)v[j / MAXV][0] | ((T)v[j / MAXV][1] << 32);
                r[j + 1] = (T)v[j / MAXV][2] | ((T)v[j / MAXV][3] << 32);
            } else {
                // An explicit move AFTER the wait: a plain copy lets hipcc satisfy the wait statement's tied operand by
                // copying the (not yet landed) destination register into the loop-carried register BEFORE the
                // s_waitcnt (seen in every u32 instance; tests/test_async_load_guard.py now checks the code objects).
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    uint32_t x;
                    asm("v_mov_b32 %0, %1" : "=v"(x) : "v"(v[j / MAXV][i]));
                    r[j + i] = (T)x;
                }
            }
        }
    }

    // position of the GI-th stage bit of mask GM in visiting order (forward: highest first)
    static constexpr int nth_stage_bit(uint32_t gm, int gi) {
        int seen = 0;
        for (int bb = 0; bb < 32; ++bb) {
            const int cand = INV ? bb : (31 - bb);
            if ((gm >> cand) & 1u) {
                if (seen == gi) return cand;
                ++seen;
            }
        }
        return 0;
    }

    // one butterfly stage (the GI-th of pass K) on the thread's registers
    // ---- LDS twiddle image (persistent kernel): every stage whose twiddle depends on the thread gets a
    // dense [h][u] slab, u = tid >> SHIFT enumerating the distinct values of (ebase >> (b+1)) ----
    struct StageGeom {
        int b, k, nhi, shift, d;
        bool uniform;
    };
    static constexpr StageGeom geom(int K, int GI) {
        const uint32_t RM = S::RMASK[K], GM = S::GMASK[K], CM = FULL & ~RM;
        const int b = nth_stage_bit(GM, GI);
        const int k = crank(RM, b);
        const int shift = cpop(CM & ((2u << b) - 1u));      // thread-id bits that sit at or below bit b
        const int dbits = cpop(CM >> (b + 1));              // thread-id bits above bit b
        return StageGeom{b, k, 1 << (LOGE - 1 - k), shift, 1 << dbits, dbits == 0};
    }
    static constexpr int img_off(int K, int GI) {  // entries before stage (K, GI)
        int off = 0;
        for (int kk = 0; kk < NPASS; ++kk)
            for (int gi = 0; gi < cpop(S::GMASK[kk]); ++gi) {
                if (kk == K && gi == GI) return off;
                const StageGeom g = geom(kk, gi);
                if (!g.uniform) off += g.nhi * g.d;
            }
        return off;
    }
    static constexpr int IMG_ENTRIES = img_off(NPASS, 0) > 0 ? img_off(NPASS, 0) : 1;
    // number of butterfly stages of the transform that precede stage (K, GI)
    static constexpr int stage_no(int K, int GI) {
        int n = GI;
        for (int kk = 0; kk < K; ++kk) n += cpop(S::GMASK[kk]);
        return n;
    }

    template <int K, int GI>
    static __device__ __forceinline__ void fill_image_stage(TwPair<T> *img, const TwPair<T> *__restrict__ tw) {
        constexpr StageGeom g = geom(K, GI);
        if constexpr (!g.uniform) {
            constexpr uint32_t RM = S::RMASK[K], CM = FULL & ~RM;
            constexpr int OFF = img_off(K, GI);
            for (uint32_t e = threadIdx.x; e < (uint32_t)(g.nhi * g.d); e += blockDim.x) {
                const uint32_t h = e / (uint32_t)g.d, u = e % (uint32_t)g.d;
                const uint32_t idx = (1u << (LOGN - 1 - g.b)) + (pdep<CM>(u << g.shift) >> (g.b + 1)) +
                                     (pdep<RM>(h << (g.k + 1)) >> (g.b + 1));
                img[OFF + e] = tw[idx];
            }
        }
    }
    template <int K = 0, int GI = 0>
    static __device__ __forceinline__ void fill_image(TwPair<T> *img, const TwPair<T> *__restrict__ tw) {
        if constexpr (K < NPASS) {
            if constexpr (GI < cpop(S::GMASK[K])) {
                fill_image_stage<K, GI>(img, tw);
                fill_image<K, GI + 1>(img, tw);
            } else {
                fill_image<K + 1, 0>(img, tw);
            }
        }
    }

    // Table entry toff + c (c: the register part of the index, a compile-time constant after unrolling) as
    //     [uniform base + 4 KiB window of c: scalar add] + [32-bit byte offset of toff: one VGPR] + [rest of c: the instruction's immediate]
    // -- the saddr form of global_load.  Written as tw + (toff + c) * 8 the constant sits inside the zero-extended product, hipcc
    // builds a 64-bit VGPR address per load (v_lshl_add_u64 / v_add_co / v_addc: ~3 VALU instructions per twiddle of every kernel
    // that reads its twiddles from global memory).
    static __device__ __forceinline__ TwPair<T> tw_at(const TwPair<T> *__restrict__ tw, uint32_t toff, uint32_t c) {
        const uint32_t cb = c * (uint32_t)sizeof(TwPair<T>), win = cb & ~4095u, imm = cb & 4095u;
        const char *base = reinterpret_cast<const char *>(tw) + win;
        return *reinterpret_cast<const TwPair<T> *>(base + (size_t)(toff * (uint32_t)sizeof(TwPair<T>)) + imm);
    }
    // one chunk of a stage's twiddles: CH consecutive values of h (the register bits above the stage bit)
    template <int K, int GI, bool IMG, int CH>
    static __device__ __forceinline__ void stage_twiddles(TwPair<T> (&w)[CH], int h0, uint32_t toff, uint32_t tid, const TwPair<T> *img,
                                                          const TwPair<T> *__restrict__ tw) {
        constexpr uint32_t RM = S::RMASK[K], GM = S::GMASK[K];
        constexpr int b = nth_stage_bit(GM, GI), k = crank(RM, b);
        constexpr StageGeom g = geom(K, GI);
#pragma unroll
        for (int hh = 0; hh < CH; ++hh) {
            const int h = h0 + hh;
            if constexpr (IMG && !g.uniform)
                w[hh] = img[img_off(K, GI) + h * g.d + (tid >> g.shift)];
            else
                w[hh] = tw_at(tw, toff, cdep((uint32_t)h << (k + 1), RM) >> (b + 1));
        }
    }
    // the butterflies of the stage whose twiddle index h lies in [h0, h0 + CH)
    template <int K, int GI, int CH, bool UNI, class R>
    static __device__ __forceinline__ void stage_butterflies(R (&r)[E], const TwPair<T> (&w)[CH], int h0, const ModParams<T> &P) {
        constexpr uint32_t RM = S::RMASK[K], GM = S::GMASK[K];
        constexpr int b = nth_stage_bit(GM, GI), k = crank(RM, b);
        constexpr bool FIRST = !INV && !SUB && stage_no(K, GI) == 0;   // inputs of a whole forward transform: canonical (Bfly::fwd)
#pragma unroll
        for (int j = 0; j < E; ++j) {
            if ((j >> k) & 1) continue;
            const int h = j >> (k + 1);
            if (h < h0 || h >= h0 + CH) continue;
            // twiddles that do not depend on the thread come from scalar loads and stay in SGPRs (never for the
            // sub-block kernels: their table prefix depends on the polynomial a thread works on)
            if constexpr (!std::is_same<R, T>::value) {
                if constexpr (INV) BoxOps<CLS>::template inv<UNI>(r[j], r[j | (1 << k)], w[h - h0].w, w[h - h0].ws, P);
                else BoxOps<CLS>::template fwd<UNI, FIRST>(r[j], r[j | (1 << k)], w[h - h0].w, w[h - h0].ws, P);
            } else if constexpr (INV)
                Bfly<T, CLS>::template inv<UNI>(r[j], r[j | (1 << k)], w[h - h0].w, w[h - h0].ws, P);
            else
                Bfly<T, CLS>::template fwd<UNI, FIRST>(r[j], r[j | (1 << k)], w[h - h0].w, w[h - h0].ws, P);
        }
    }

    // NORM (inverse only): the stage on the top index bit -- the last one, one twiddle inv_twid[1] for all of its
    // butterflies -- also applies the 1/N normalisation (Bfly::inv_norm).
    // R: the register form of a coefficient -- T itself, or (32-bit lazy class) a 64-bit "box" whose low word is the
    // value and whose high word is don't-care, the form in which x + y w + q (-p) is two v_mad_u64_u32 (BoxOps).
    template <int K, int GI, bool IMG = false, bool NORM = false, int TWC = 0, class R = T>
    static __device__ __forceinline__ void stage(R (&r)[E], uint32_t ebase, uint32_t qpre, uint32_t depth,
                                                 const TwPair<T> *__restrict__ tw, const ModParams<T> &P,
                                                 uint32_t tid = 0, const TwPair<T> *img = nullptr) {
        constexpr uint32_t RM = S::RMASK[K], GM = S::GMASK[K], CM = FULL & ~RM;
        constexpr int b = nth_stage_bit(GM, GI);  // element-index bit of this stage
        constexpr int k = crank(RM, b);           // register-index bit that carries it
        constexpr int NHI = 1 << (LOGE - 1 - k);  // register bits above it -> distinct twiddles per thread
        // table entry = 2^(LOGN-1-b) + (e >> (b+1))  [<< depth / + sub-block prefix for SUB kernels]
        uint32_t toff = (1u << (LOGN - 1 - b)) << (SUB ? depth : 0u);
        if constexpr ((CM >> (b + 1)) != 0u) toff += ebase >> (b + 1);
        if constexpr (SUB) toff += qpre >> (b + 1);
        if constexpr (NORM && INV && b == LOGN - 1) {
            static_assert(!SUB, "the normalising stage is the last stage of a whole transform");
#pragma unroll
            for (int j = 0; j < E; ++j) {
                if ((j >> k) & 1) continue;
                if constexpr (std::is_same<R, T>::value) Bfly<T, CLS>::inv_norm(r[j], r[j | (1 << k)], P);
                else BoxOps<CLS>::inv_norm(r[j], r[j | (1 << k)], P);
            }
            return;
        }
        // Twiddles of the stage, TWC at a time (TWC = 0: all NHI at once): bounds the registers the thread-dependent
        // twiddle loads hold (global-memory twiddles of the large sizes; workgroup shapes compiled for 128 VGPRs).
        constexpr StageGeom g = geom(K, GI);
        // FAM 2 (wave blocks): the thread-id bits above the stage bit are wavefront-index bits only (the six lane bits
        // sit at or below it), so the twiddle index is wave-uniform: scalar loads, SGPR operands.
        // (FAM 2 with SUB -- the two 16384-point halves of a 32768-point transform, ntt_blk.hpp -- keeps both: there the
        // sub-block prefix is a workgroup-uniform constant, not a per-thread value)
        constexpr bool WAVE_UNI = FAM == 2 && !g.uniform && g.shift >= 6 && TPP >= 64;
        if constexpr (WAVE_UNI || (FAM == 2 && SUB && g.uniform)) toff = (uint32_t)__builtin_amdgcn_readfirstlane((int)toff);
        constexpr bool UNI = (g.uniform && (!SUB || FAM == 2)) || WAVE_UNI;
        constexpr bool FIRST = !INV && !SUB && stage_no(K, GI) == 0;   // inputs of a whole forward transform: canonical (Bfly::fwd)
        constexpr bool CHUNKED = TWC > 0 && !UNI && NHI > TWC;
        constexpr int CH = CHUNKED ? TWC : NHI;
        if constexpr (CHUNKED && FAM == 2) {
            // Double-buffered chunks (wave-block kernels): the loads of chunk c + 1 are in flight behind the butterflies of
            // chunk c.  The waves of a workgroup run these passes in step behind their barrier, so a load waited for right
            // after its issue idles the whole SIMD for an L2 round trip, once per chunk.
            TwPair<T> wa[CH], wb[CH];
            stage_twiddles<K, GI, IMG, CH>(wa, 0, toff, tid, img, tw);
#pragma unroll
            for (int h0 = 0; h0 < NHI; h0 += 2 * CH) {
                if (h0 + CH < NHI) stage_twiddles<K, GI, IMG, CH>(wb, h0 + CH, toff, tid, img, tw);
                __builtin_amdgcn_sched_barrier(0);
                stage_butterflies<K, GI, CH, UNI, R>(r, wa, h0, P);
                __builtin_amdgcn_sched_barrier(0);
                if (h0 + CH < NHI) {
                    if (h0 + 2 * CH < NHI) stage_twiddles<K, GI, IMG, CH>(wa, h0 + 2 * CH, toff, tid, img, tw);
                    __builtin_amdgcn_sched_barrier(0);
                    stage_butterflies<K, GI, CH, UNI, R>(r, wb, h0 + CH, P);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        } else {
#pragma unroll
            for (int h0 = 0; h0 < NHI; h0 += CH) {
                TwPair<T> w[CH];
#pragma unroll
                for (int hh = 0; hh < CH; ++hh) {
                    const int h = h0 + hh;
                    if constexpr (IMG && !g.uniform)
                        w[hh] = img[img_off(K, GI) + h * g.d + (tid >> g.shift)];
                    else
                        w[hh] = tw_at(tw, toff, cdep((uint32_t)h << (k + 1), RM) >> (b + 1));
                }
#pragma unroll
                for (int j = 0; j < E; ++j) {
                    if ((j >> k) & 1) continue;
                    const int h = j >> (k + 1);
                    if (h < h0 || h >= h0 + CH) continue;
                    // twiddles that do not depend on the thread come from scalar loads and stay in SGPRs (never for the
                    // sub-block kernels: their table prefix depends on the polynomial a thread works on)
                    if constexpr (!std::is_same<R, T>::value) {
                        if constexpr (INV) BoxOps<CLS>::template inv<UNI>(r[j], r[j | (1 << k)], w[h - h0].w, w[h - h0].ws, P);
                        else BoxOps<CLS>::template fwd<UNI, FIRST>(r[j], r[j | (1 << k)], w[h - h0].w, w[h - h0].ws, P);
                    } else if constexpr (INV)
                        Bfly<T, CLS>::template inv<UNI>(r[j], r[j | (1 << k)], w[h - h0].w, w[h - h0].ws, P);
                    else
                        Bfly<T, CLS>::template fwd<UNI, FIRST>(r[j], r[j | (1 << k)], w[h - h0].w, w[h - h0].ws, P);
                }
                if constexpr (CHUNKED) __builtin_amdgcn_sched_barrier(0);  // keep the chunks (and their registers) apart
            }
        }
        if constexpr (Bfly<T, CLS>::IS_FP) {
            // range reductions of the double-held residues (bounds and periods: BflyFp in ntt_arith.hpp).  The last
            // stage is followed by finish_*.
            static_assert(!SUB || FAM == 2, "the double-precision classes cover whole LDS-resident transforms (and the halves of Ntt32k, which reduces around its own stage)");
            using BF = Bfly<T, CLS>;
            constexpr int SNO = stage_no(K, GI);
            if constexpr (!INV && SNO % BF::FWD_REDUCE_EVERY == BF::FWD_REDUCE_EVERY - 1 && SNO != LOGN - 1) {
#pragma unroll
                for (int j = 0; j < E; ++j) r[j] = BF::reduce(r[j], P);
            }
            if constexpr (INV && SNO % BF::INV_REDUCE_EVERY == BF::INV_REDUCE_EVERY - 1 && SNO != LOGN - 1) {
#pragma unroll
                for (int j = 0; j < E; ++j)
                    if (((j >> k) & 1) == 0) r[j] = BF::reduce(r[j], P);
            }
        }
    }

    template <int K, int GI, bool IMG, bool NORM, int TWC, class R>
    static __device__ __forceinline__ void stages_r(R (&r)[E], uint32_t ebase, uint32_t qpre, uint32_t depth,
                                                    const TwPair<T> *__restrict__ tw, const ModParams<T> &P, uint32_t tid,
                                                    const TwPair<T> *img) {
        if constexpr (GI < cpop(S::GMASK[K])) {
            stage<K, GI, IMG, NORM, TWC, R>(r, ebase, qpre, depth, tw, P, tid, img);
            stages_r<K, GI + 1, IMG, NORM, TWC, R>(r, ebase, qpre, depth, tw, P, tid, img);
        }
    }
    // all register-resident stages of pass K
    template <int K, int GI = 0, bool IMG = false, bool NORM = false, int TWC = 0>
    static __device__ __forceinline__ void stages(T (&r)[E], uint32_t ebase, uint32_t qpre, uint32_t depth,
                                                  const TwPair<T> *__restrict__ tw, const ModParams<T> &P,
                                                  uint32_t tid = 0, const TwPair<T> *img = nullptr) {
        if constexpr (BoxOps<CLS>::template USE<T>::value) {
            uint64_t c[E];
#pragma unroll
            for (int j = 0; j < E; ++j) c[j] = BoxOps<CLS>::box((uint32_t)r[j], P);
            stages_r<K, GI, IMG, NORM, TWC, uint64_t>(c, ebase, qpre, depth, tw, P, tid, img);
#pragma unroll
            for (int j = 0; j < E; ++j) r[j] = (T)BoxOps<CLS>::unbox(c[j], P);
        } else {
            stages_r<K, GI, IMG, NORM, TWC, T>(r, ebase, qpre, depth, tw, P, tid, img);
        }
    }

    template <int K>
    static __device__ __forceinline__ void run_pass(T (&r)[E], T *__restrict__ g, T *lds, uint32_t tid, bool active,
                                                    uint32_t qpre, uint32_t depth, const TwPair<T> *__restrict__ tw,
                                                    const ModParams<T> &P) {
        constexpr uint32_t RM = S::RMASK[K], CM = FULL & ~RM;
        const uint32_t ebase = pdep<CM>(tid);
        if constexpr (K == 0) {
            if (active) gather<RM>(r, (const T *)g, ebase, false);
#pragma unroll
            for (int j = 0; j < E; ++j) r[j] = Bfly<T, CLS>::load_fix(r[j], P);
        } else {
            gather<RM>(r, (const T *)lds, ebase, true);
        }
        stages<K>(r, ebase, qpre, depth, tw, P);
        if constexpr (K == NPASS - 1) {
#pragma unroll
            for (int j = 0; j < E; ++j) r[j] = INV ? Bfly<T, CLS>::finish_inv(r[j], P) : Bfly<T, CLS>::finish_fwd(r[j], P);
            if (active) scatter<RM>(r, g, ebase, false);
        } else {
            if constexpr (K > 0) sync();  // everyone has read the previous exchange before it is overwritten
            scatter<RM>(r, lds, ebase, true);
            sync();
            run_pass<K + 1>(r, g, lds, tid, active, qpre, depth, tw, P);
        }
    }

    static __device__ __forceinline__ void run(T *__restrict__ data, const TwPair<T> *__restrict__ tw,
                                               const ModParams<T> &P, uint32_t nsub, uint32_t depth, T *lds_all) {
        const uint32_t tid = threadIdx.x & (TPP - 1);
        const uint32_t pl = threadIdx.x / TPP;
        const uint32_t sub = blockIdx.x * PPB + pl;
        const bool active = sub < nsub;
        T *g = data + ((size_t)sub << LOGN);
        T *lds = lds_all + (size_t)pl * LDS_WORDS_1;
        uint32_t qpre = 0;
        if constexpr (SUB) qpre = (sub & ((1u << depth) - 1u)) << LOGN;
        T r[E];
#pragma unroll
        for (int j = 0; j < E; ++j) r[j] = 0;
        run_pass<0>(r, g, lds, tid, active, qpre, depth, tw, P);
    }
};

// FAM: schedule family (gen_sched.py).  32-bit words from N = 8192 up run family 3 (padded exchange layout: one LDS base per pass +
// immediate offsets instead of a shift / xor / or per access -- a sixth of the instructions of these VALU-bound kernels).
template <class T, int LOGN, bool SUB> constexpr int plain_fam() { return (sizeof(T) == 4 && LOGN >= 13 && !SUB) ? 3 : 0; }

// -------------------------------------------------------------------------------------------------
// Persistent, software-pipelined variant for polynomials that live in one wavefront (N <= 1024).
// Each workgroup stages the thread-dependent twiddles once in LDS (fill_image) and then walks tiles of
// WP_PPB polynomials: the global loads of tile i+1 are issued before the butterflies of tile i, and no
// vector-memory wait sits inside the compute phase (twiddles come from LDS / scalar loads), so HBM
// streaming overlaps the VALU-bound passes instead of alternating with them.
// -------------------------------------------------------------------------------------------------
template <class T, int LOGN, bool INV, int CLS, int WPB, int FAM = 0>
struct NttWp : NttKernel<T, LOGN, INV, CLS, false, FAM> {
    using B = NttKernel<T, LOGN, INV, CLS, false, FAM>;
    using S = typename B::S;
    static constexpr int E = B::E, TPP = B::TPP, NPASS = B::NPASS;
    static constexpr int BLOCK = WPB, PPB = WPB / TPP;
    static constexpr uint32_t FULL = B::FULL;
    static_assert(NPASS > 1 && TPP <= WPB, "multi-pass transforms owned by at most one workgroup");

    // exchange synchronisation.  Inside one wavefront: compiler barrier only (B::sync).  Across the waves of a
    // workgroup: raw s_barrier behind an LDS-only wait -- __syncthreads() would also make hipcc drain vmcnt(0),
    // i.e. wait for the prefetched global loads of the next tile at every exchange.
    static __device__ __forceinline__ void wsync() {
        if constexpr (B::WAVE_PRIVATE) {
            B::sync();
        } else {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
        }
    }


    // IMG: thread-dependent twiddles come from the workgroup's LDS image (fill_image); otherwise from the table in
    // global memory (L2), for kernels that walk several primes and cannot hold an image per prime
    // FIN: bring the outputs into [0, p) (what memory holds); fused kernels whose next step takes the class's lazy
    // register form (CLS_FP products) skip it
    template <int K, bool NORM = false, bool IMG = true, bool FIN = true, int TWC = 0>
    static __device__ __forceinline__ void pass(T (&r)[E], T *lds, uint32_t tid, const TwPair<T> *__restrict__ tw,
                                                const TwPair<T> *img, const ModParams<T> &P) {
        constexpr uint32_t RM = S::RMASK[K], CM = FULL & ~RM;
        const uint32_t ebase = pdep<CM>(tid);
        if constexpr (K > 0) B::template gather<RM>(r, (const T *)lds, ebase, true);
        B::template stages<K, 0, IMG, NORM, TWC>(r, ebase, 0u, 0u, tw, P, tid, img);
        if constexpr (K < NPASS - 1) {
            if constexpr (K > 0) wsync();
            B::template scatter<RM>(r, lds, ebase, true);
            wsync();
            pass<K + 1, NORM, IMG, FIN, TWC>(r, lds, tid, tw, img, P);
        } else if constexpr (FIN) {
#pragma unroll
            for (int j = 0; j < E; ++j) r[j] = INV ? Bfly<T, CLS>::finish_inv(r[j], P) : Bfly<T, CLS>::finish_fwd(r[j], P);
        }
    }

    // HBM is always accessed in the fully coalesced layout IO_RM (16 bytes per lane, consecutive lanes on
    // consecutive addresses: 1 KiB per wave-instruction); when the first / last pass wants another register
    // layout the tile takes one extra LDS transpose instead of 16-byte accesses at a 32-128 byte lane stride.
    static constexpr int VB = (B::MAXV == 4) ? 2 : 1;
    static constexpr uint32_t IO_RM = ((1u << VB) - 1u) | (((1u << (B::LOGE - VB)) - 1u) << (LOGN - (B::LOGE - VB)));

    // USE_IMG = false: no LDS twiddle image (sizes whose image does not fit next to the exchange buffer): the
    // thread-dependent twiddles are read from the table in global memory (L2) inside the passes.
    template <bool USE_IMG = true, int TWC_IMG = 0>
    static __device__ __forceinline__ void run(T *__restrict__ data, const TwPair<T> *__restrict__ tw,
                                               const ModParams<T> &P, uint32_t nsub, T *lds_all, TwPair<T> *img) {
        if constexpr (USE_IMG) {
            B::fill_image(img, tw);
            __syncthreads();
        }
        const uint32_t tid = threadIdx.x & (TPP - 1);
        const uint32_t pl = threadIdx.x / TPP;
        const bool nt = P.stream != 0;   // workgroup-uniform: the batch streams (see gather_tile / NT above)
        T *lds = lds_all + (size_t)pl * B::LDS_WORDS_1;
        constexpr uint32_t RM0 = S::RMASK[0], CM0 = FULL & ~RM0;
        constexpr uint32_t RML = S::RMASK[NPASS - 1], CML = FULL & ~RML;
        constexpr uint32_t CMIO = FULL & ~IO_RM;
        const uint32_t ebase0 = pdep<CM0>(tid), ebaseL = pdep<CML>(tid), ebaseIO = pdep<CMIO>(tid);
        const uint32_t ntiles = (nsub + PPB - 1) / PPB;
        // byte offset of this thread's first vector inside a tile of PPB polynomials (IO layout)
        const uint32_t voffIO = ((pl << LOGN) + ebaseIO) * (uint32_t)sizeof(T);
        T r[E];
#pragma unroll
        for (int j = 0; j < E; ++j) r[j] = 0;
        uint32_t tile = blockIdx.x;
        // The first tile takes the same asynchronous path as every later one: plain C++ loads here leave hipcc's vmcnt
        // model with loads in flight at the loop header, and the waits it then places INSIDE the loop drain the next
        // tile's prefetch right after it was issued on every later iteration (no overlap of HBM and butterflies;
        // tests/test_async_load_guard.py now measures the distance between every prefetch and its retiring wait).
        if (tile < ntiles) {
            const uint32_t last = nsub - 1u - tile * PPB;   // ragged tail: clamp to the last polynomial (in range)
            const uint32_t pl0 = pl < last ? pl : last;
            typename B::AsyncVec v0[E / B::MAXV];
            B::template gather_async_rt<IO_RM>(v0, (const T *)(data + (((size_t)tile * PPB) << LOGN)),
                                               ((pl0 << LOGN) + ebaseIO) * (uint32_t)sizeof(T), nt);
            B::template wait_async<0>(v0);
            B::unpack_async(r, v0);
        }
        constexpr int NST = E / B::template vec_elems<IO_RM>();  // store instructions per tile (younger than the prefetch)
        for (; tile < ntiles; tile += gridDim.x) {
            const uint32_t sub = tile * PPB + pl;
            const uint32_t tnext = tile + gridDim.x;
            const bool more = tnext < ntiles;  // workgroup-uniform
            T *tbase = data + (((size_t)tile * PPB) << LOGN);
            // In the shapes compiled for 128 VGPRs every address of the tile body (layout offsets, twiddle offsets) is
            // recomputed per tile from an opaque copy of the thread index: left to itself hipcc hoists all of them out
            // of the tile loop and then spills the lot.
            constexpr bool RECOMPUTE = !USE_IMG || TWC_IMG > 0;  // the shapes compiled for 128 VGPRs
            uint32_t tidv = tid;
            if constexpr (RECOMPUTE) asm volatile("" : "+v"(tidv));
            const uint32_t eb0 = RECOMPUTE ? pdep<CM0>(tidv) : ebase0, ebL = RECOMPUTE ? pdep<CML>(tidv) : ebaseL;
            const uint32_t ebIO = RECOMPUTE ? pdep<CMIO>(tidv) : ebaseIO;
            const uint32_t voff = RECOMPUTE ? ((pl << LOGN) + ebIO) * (uint32_t)sizeof(T) : voffIO;
            typename B::AsyncVec vn[E / B::MAXV];
            if (more) {
                // ragged tail: lanes of polynomials past the end re-read the last polynomial (harmless, in range)
                const uint32_t last = nsub - 1u - tnext * PPB;
                const uint32_t pln = pl < last ? pl : last;
                B::template gather_async_rt<IO_RM>(vn, (const T *)(data + (((size_t)tnext * PPB) << LOGN)),
                                                   ((pln << LOGN) + ebIO) * (uint32_t)sizeof(T), nt);
            }
#pragma unroll
            for (int j = 0; j < E; ++j) r[j] = Bfly<T, CLS>::load_fix(r[j], P);  // memory word -> register form
            if constexpr (RM0 != IO_RM) {  // input transpose
                B::template scatter<IO_RM>(r, lds, ebIO, true);
                wsync();
                B::template gather<RM0>(r, (const T *)lds, eb0, true);
                wsync();
            }
            pass<0, false, USE_IMG, true, USE_IMG ? TWC_IMG : 2>(r, lds, tidv, tw, img, P);
            if constexpr (RML != IO_RM) {  // output transpose
                wsync();
                B::template scatter<RML>(r, lds, ebL, true);
                wsync();
                B::template gather<IO_RM>(r, (const T *)lds, ebIO, true);
            }
            if (sub < nsub) B::template scatter_tile_rt<IO_RM>(r, tbase, voff, nt);
            wsync();  // the exchange buffer is reused by the next tile
            if (more) {
                // lanes of inactive polynomials skipped their stores: the counter then allows fewer
                // outstanding operations than were issued, i.e. the wait is only stricter.
                B::template wait_async<NST>(vn);
                B::unpack_async(r, vn);
            }
        }
    }
};

template <class T, int LOGN, bool INV, int CLS, int WPB, int WPW>
__global__ __launch_bounds__(WPB, WPW) void ntt_kernel_wp(T *__restrict__ data, const TwPair<T> *__restrict__ tw,
                                                      const ModParams<T> P, uint32_t nsub) {
    using K = NttWp<T, LOGN, INV, CLS, WPB>;
    __shared__ __attribute__((aligned(16))) T lds[(size_t)K::PPB * K::B::LDS_WORDS_1];
    __shared__ __attribute__((aligned(16))) TwPair<T> img[K::B::IMG_ENTRIES];
    K::run(data, tw, P, nsub, lds, img);
}

// -------------------------------------------------------------------------------------------------
// Fused negacyclic product against a pre-transformed operand (the K1 -> K3 -> K2 pipeline of SURVEY 2.1):
//     lhs <- inv( mul_assign_normalize( fwd(lhs), rhs_ntt ) )
// i.e. what a caller of the reference writes as plan.fwd(a); plan.mul_assign_normalize(a, b_ntt); plan.inv(a)
// (examples/mul_poly_prime.rs), in ONE pass over HBM: 3*N words of traffic instead of 7*N.  The forward
// transform's last register layout is the inverse transform's first one (mirror schedules), so the
// pointwise product happens in registers between the two and rhs_ntt is read in that layout.
// -------------------------------------------------------------------------------------------------
template <class T, int LOGN, int CLS, int WPB>
struct MulWp {
    using F = NttWp<T, LOGN, false, CLS, WPB>;
    using I = NttWp<T, LOGN, true, CLS, WPB>;
    using FB = typename F::B;
    using IB = typename I::B;
    static constexpr int E = FB::E, TPP = FB::TPP, NPASS = FB::NPASS, PPB = WPB / TPP;
    static constexpr uint32_t FULL = FB::FULL;
    static constexpr uint32_t RM0 = FB::S::RMASK[0], RMM = FB::S::RMASK[NPASS - 1], RML = IB::S::RMASK[NPASS - 1];
    static_assert(RMM == IB::S::RMASK[0], "forward and inverse schedules must mirror each other");
    static constexpr uint32_t IO_RM = F::IO_RM;

    static __device__ __forceinline__ void run(T *__restrict__ lhs, const T *__restrict__ rhs_ntt,
                                               const TwPair<T> *__restrict__ twf, const TwPair<T> *__restrict__ twi,
                                               const ModParams<T> &P, uint32_t nsub, T *lds_all, TwPair<T> *imgf,
                                               TwPair<T> *imgi) {
        FB::fill_image(imgf, twf);
        IB::fill_image(imgi, twi);
        __syncthreads();
        const ModParams<T> Pi = mul_inv_params<T, CLS>(P);   // the inverse half's constants (lazy class: times 2^B, see mul_fused)
        const uint32_t tid = threadIdx.x & (TPP - 1);
        const uint32_t pl = threadIdx.x / TPP;
        T *lds = lds_all + (size_t)pl * FB::LDS_WORDS_1;
        constexpr uint32_t CM0 = FULL & ~RM0, CMM = FULL & ~RMM, CML = FULL & ~RML, CMIO = FULL & ~IO_RM;
        const uint32_t ebase0 = pdep<CM0>(tid), ebaseM = pdep<CMM>(tid), ebaseL = pdep<CML>(tid), ebaseIO = pdep<CMIO>(tid);
        const uint32_t ntiles = (nsub + PPB - 1) / PPB;
        const uint32_t voffIO = ((pl << LOGN) + ebaseIO) * (uint32_t)sizeof(T);
        T r[E];
#pragma unroll
        for (int j = 0; j < E; ++j) r[j] = 0;
        uint32_t tile = blockIdx.x;
        if (tile < ntiles) {  // asynchronous like every later tile (see NttWp::run)
            const uint32_t last = nsub - 1u - tile * PPB;
            const uint32_t pl0 = pl < last ? pl : last;
            typename FB::AsyncVec v0[E / FB::MAXV];
            FB::template gather_async<IO_RM>(v0, (const T *)(lhs + (((size_t)tile * PPB) << LOGN)),
                                             ((pl0 << LOGN) + ebaseIO) * (uint32_t)sizeof(T));
            FB::template wait_async<0>(v0);
            FB::unpack_async(r, v0);
        }
        constexpr int NST = E / FB::template vec_elems<IO_RM>();
        for (; tile < ntiles; tile += gridDim.x) {
            const uint32_t sub = tile * PPB + pl;
            const uint32_t tnext = tile + gridDim.x;
            const bool more = tnext < ntiles;
            T *tbase = lhs + (((size_t)tile * PPB) << LOGN);
            typename FB::AsyncVec vn[E / FB::MAXV];
            if (more) {
                const uint32_t last = nsub - 1u - tnext * PPB;
                const uint32_t pln = pl < last ? pl : last;
                FB::template gather_async<IO_RM>(vn, (const T *)(lhs + (((size_t)tnext * PPB) << LOGN)),
                                                 ((pln << LOGN) + ebaseIO) * (uint32_t)sizeof(T));
            }
#pragma unroll
            for (int j = 0; j < E; ++j) r[j] = Bfly<T, CLS>::load_fix(r[j], P);
            if constexpr (RM0 != IO_RM) {
                FB::template scatter<IO_RM>(r, lds, ebaseIO, true);
                F::wsync();
                FB::template gather<RM0>(r, (const T *)lds, ebase0, true);
                F::wsync();
            }
            // NTT-domain values in layout RMM: canonical, or (CLS_FP) the lazy doubles mul_for_inv takes
            F::template pass<0, false, true, mul_fwd_fin<T, CLS>()>(r, lds, tid, twf, imgf, P);
            {
                // clamped polynomial index for the reads of a ragged tail
                const uint32_t lastc = nsub - 1u - tile * PPB;
                const uint32_t plc = pl < lastc ? pl : lastc;
                T b[E];
                FB::template gather_tile<RMM>(b, rhs_ntt + (((size_t)tile * PPB) << LOGN),
                                              ((plc << LOGN) + ebaseM) * (uint32_t)sizeof(T));
#pragma unroll
                for (int j = 0; j < E; ++j) r[j] = mul_fused<T, CLS>(r[j], b[j], P);  // 1/N: inside the last inverse stage
            }
            F::wsync();  // the forward transform's last exchange has been read before the inverse overwrites it
            I::template pass<0, true>(r, lds, tid, twi, imgi, Pi);  // canonical coefficients, layout RML
            if constexpr (RML != IO_RM) {
                F::wsync();
                FB::template scatter<RML>(r, lds, ebaseL, true);
                F::wsync();
                FB::template gather<IO_RM>(r, (const T *)lds, ebaseIO, true);
            }
            if (sub < nsub) FB::template scatter_tile<IO_RM>(r, tbase, voffIO);
            F::wsync();
            if (more) {
                FB::template wait_async<NST>(vn);
                FB::unpack_async(r, vn);
            }
        }
    }
};

template <class T, int LOGN, int CLS, int WPB, int WPW>
__global__ __launch_bounds__(WPB, WPW) void mul_kernel_wp(T *__restrict__ lhs, const T *__restrict__ rhs_ntt,
                                                      const TwPair<T> *__restrict__ twf,
                                                      const TwPair<T> *__restrict__ twi, const ModParams<T> P,
                                                      uint32_t nsub) {
    using K = MulWp<T, LOGN, CLS, WPB>;
    __shared__ __attribute__((aligned(16))) T lds[(size_t)K::PPB * K::FB::LDS_WORDS_1];
    __shared__ __attribute__((aligned(16))) TwPair<T> imgf[K::FB::IMG_ENTRIES];
    __shared__ __attribute__((aligned(16))) TwPair<T> imgi[K::IB::IMG_ENTRIES];
    K::run(lhs, rhs_ntt, twf, twi, P, nsub, lds, imgf, imgi);
}

// The same fused product for the sizes that have no LDS twiddle image (32-bit words, N = 8192 ... 32768): one polynomial per
// workgroup, twiddles from the tables in global memory (L2), no persistent walk -- several workgroups per CU in different
// phases overlap loads, butterflies and stores, as for the stand-alone transforms of these sizes.  3 N words of HBM traffic
// instead of 7 N and one launch instead of three.
template <class T, int LOGN, int CLS>
struct MulOne {
    static constexpr int TPP = NttKernel<T, LOGN, false, CLS, false>::TPP;
    static constexpr int FAM = plain_fam<T, LOGN, false>();
    using F = NttWp<T, LOGN, false, CLS, TPP, FAM>;
    using I = NttWp<T, LOGN, true, CLS, TPP, FAM>;
    using FB = typename F::B;
    using IB = typename I::B;
    static constexpr int E = FB::E, NPASS = FB::NPASS;
    static constexpr uint32_t FULL = FB::FULL;
    static constexpr uint32_t RM0 = FB::S::RMASK[0], RMM = FB::S::RMASK[NPASS - 1], RML = IB::S::RMASK[NPASS - 1];
    static_assert(RMM == IB::S::RMASK[0], "forward and inverse schedules must mirror each other");

    static __device__ __forceinline__ void run(T *__restrict__ lhs, const T *__restrict__ rhs_ntt,
                                               const TwPair<T> *__restrict__ twf, const TwPair<T> *__restrict__ twi,
                                               const ModParams<T> &P, T *lds) {
        constexpr uint32_t CM0 = FULL & ~RM0, CMM = FULL & ~RMM, CML = FULL & ~RML;
        const uint32_t tid = threadIdx.x;
        T *base = lhs + ((size_t)blockIdx.x << LOGN);
        T r[E];
        FB::template gather<RM0>(r, (const T *)base, pdep<CM0>(tid), false);
#pragma unroll
        for (int j = 0; j < E; ++j) r[j] = Bfly<T, CLS>::load_fix(r[j], P);
        // NTT-domain values in layout RMM: canonical, or the lazy form mul_for_inv takes
        F::template pass<0, false, false, mul_fwd_fin<T, CLS>()>(r, lds, tid, twf, nullptr, P);
        {
            T b[E];
            FB::template gather_tile<RMM>(b, rhs_ntt + ((size_t)blockIdx.x << LOGN), pdep<CMM>(tid) * (uint32_t)sizeof(T));
#pragma unroll
            for (int j = 0; j < E; ++j) r[j] = mul_fused<T, CLS>(r[j], b[j], P);  // 1/N: inside the last inverse stage
        }
        F::wsync();  // the forward transform's last exchange has been read before the inverse overwrites it
        uint32_t ti = tid;   // fresh opaque copy: no address of the forward half stays live into the inverse half
        asm volatile("" : "+v"(ti));
        I::template pass<0, true, false>(r, lds, ti, twi, nullptr, mul_inv_params<T, CLS>(P));  // canonical coefficients, layout RML
        FB::template scatter<RML>(r, base, pdep<CML>(ti), false);
    }
};

template <class T, int LOGN, int CLS>
__global__ __launch_bounds__((MulOne<T, LOGN, CLS>::TPP)) void mul_kernel_one(T *__restrict__ lhs, const T *__restrict__ rhs_ntt,
                                                                           const TwPair<T> *__restrict__ twf,
                                                                           const TwPair<T> *__restrict__ twi,
                                                                           const ModParams<T> P) {
    using K = MulOne<T, LOGN, CLS>;
    __shared__ __attribute__((aligned(16))) T lds[K::FB::LDS_WORDS_1];
    K::run(lhs, rhs_ntt, twf, twi, P, lds);
}

// -------------------------------------------------------------------------------------------------
// Fused mul_accumulate chain (the step around the NTT in the reference's caller):
//     for each o < NOUT:  out[b][o] (+)= inv( sum_{j < J} fwd(terms[b][j]) (.) key_ntt[j][o] )
// (`ostride` >= NOUT: outputs per batch element in `out` / per term in `key_ntt` -- a caller that splits the outputs of one chain over
// two launches passes base pointers offset to its first output and the full count as stride)
// i.e. what a caller of the reference writes as
//     for j { plan.fwd(t_j); for o { plan.mul_accumulate(acc_o, t_j, key[j][o]) } }  for o { plan.inv(acc_o) }
// (src/prime64.rs:794, :1085-1128, :872) with every intermediate kept in registers: (J + NOUT) * N words of HBM
// traffic per batch element instead of (2J + 3*J*NOUT + 2*NOUT) * N.  `key_ntt` (J * NOUT polynomials, shared
// by the whole batch) is read in the forward transform's last register layout and stays L2-resident.
// -------------------------------------------------------------------------------------------------
template <int V> struct IntC { static constexpr int value = V; };
template <int I, int N, class F> __device__ __forceinline__ void static_for(F &&f) {
    if constexpr (I < N) {
        f(IntC<I>{});
        static_for<I + 1, N>(f);
    }
}

template <class T, int LOGN, int CLS, int WPB, int NOUT, int FAM = 0>
struct ExtWp {
    using F = NttWp<T, LOGN, false, CLS, WPB, FAM>;
    using I = NttWp<T, LOGN, true, CLS, WPB, FAM>;
    using FB = typename F::B;
    using IB = typename I::B;
    static constexpr int E = FB::E, TPP = FB::TPP, NPASS = FB::NPASS, PPB = WPB / TPP;
    static constexpr uint32_t FULL = FB::FULL;
    static constexpr uint32_t RM0 = FB::S::RMASK[0], RMM = FB::S::RMASK[NPASS - 1], RML = IB::S::RMASK[NPASS - 1];
    static_assert(RMM == IB::S::RMASK[0], "forward and inverse schedules must mirror each other");
    static constexpr uint32_t IO_RM = F::IO_RM;

    // acc[e] <- acc[e] + r[e] * key[e] with the key polynomial read from global memory in layout RMM, one
    // 16-byte vector at a time (no E-element staging array: the accumulators already fill the register file)
    static __device__ __forceinline__ void mul_acc_key(T (&acc)[E], const T (&r)[E], const T *__restrict__ key,
                                                       uint32_t ebase, const ModParams<T> &P) {
        constexpr int NV = FB::template vec_elems<RMM>();
        using V = typename VecOf<T, NV>::type;
#pragma unroll
        for (int j = 0; j < E; j += NV) {
            const uint32_t e = ebase | cdep((uint32_t)j, RMM);
            if constexpr (NV == 1) {
                acc[j] = mul_acc_cls<T, CLS>(acc[j], r[j], key[e], P);
            } else {
                const V v = *reinterpret_cast<const V *>(&key[e]);
#pragma unroll
                for (int i = 0; i < NV; ++i) acc[j + i] = mul_acc_cls<T, CLS>(acc[j + i], r[j + i], v[i], P);
            }
        }
    }

    // The next term is prefetched with ordinary loads (not NttKernel::gather_async): this kernel's register
    // footprint can make hipcc spill, and an asynchronous inline-asm load into a register that the compiler
    // spills or reassigns before the data lands corrupts whatever lives there by then.
    static __device__ __forceinline__ void run(T *__restrict__ out, const T *__restrict__ terms,
                                               const T *__restrict__ key_ntt, const TwPair<T> *__restrict__ twf,
                                               const TwPair<T> *__restrict__ twi, const ModParams<T> &P, uint32_t nb,
                                               uint32_t nterms, bool accumulate, uint32_t ostride, T *lds_all, TwPair<T> *imgf,
                                               TwPair<T> *imgi) {
        FB::fill_image(imgf, twf);
        IB::fill_image(imgi, twi);
        __syncthreads();
        const uint32_t tid = threadIdx.x & (TPP - 1);
        const uint32_t pl = threadIdx.x / TPP;
        T *lds = lds_all + (size_t)pl * FB::LDS_WORDS_1;
        constexpr uint32_t CM0 = FULL & ~RM0, CMM = FULL & ~RMM, CML = FULL & ~RML, CMIO = FULL & ~IO_RM;
        const uint32_t ebase0 = pdep<CM0>(tid), ebaseM = pdep<CMM>(tid), ebaseL = pdep<CML>(tid), ebaseIO = pdep<CMIO>(tid);
        const uint32_t ntiles = (nb + PPB - 1) / PPB;
        // NEXT: the next term is loaded while the current one is transformed.  With four 64-bit accumulator tiles the
        // register file has no room for it (hipcc spilled 100 ... 270 registers): those shapes load each term where it is
        // used and recompute their layout / image offsets per term from an opaque copy of the thread index (hoisted out
        // of the term loop they stay live next to the accumulators).  Measured with tools/ext_bench.py (J = 6, N = 512 ...
        // 2048): four outputs -22 ... -45 % in every class; three outputs -18 ... -23 % for p = 2^64 - c, -10 ... -13 % for
        // the 63-bit class at N = 2048, but +9 ... +19 % for it below and +4 ... +7 % for the double-precision
        // classes, which keep the prefetch there.  The 62-bit class with three outputs, re-measured in round 4 on the Montgomery
        // accumulate (whose 128-bit intermediates made the prefetching shape spill 41 ... 57 registers): -13 ... -24 % at
        // N = 256 ... 1024, -5 % at N = 128, +-1 % at N = 64, +6 % at N = 32 (which keeps the prefetch).
        constexpr bool NEXT = !(sizeof(T) == 8 &&
                                (NOUT == 4 || (NOUT == 3 && (CLS == CLS_PM64 || CLS == CLS_GENERIC || (LOGN >= 7 && CLS == CLS_LAZY) ||
                                                             (LOGN >= 11 && CLS == CLS_STRICT)))));
        constexpr bool OPAQUE = !NEXT;   // (with the prefetch kept, recomputing per term measured +-0 ... +6 %: not used there)
        for (uint32_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
            const uint32_t b = tile * PPB + pl;
            const uint32_t bc = b < nb ? b : nb - 1;  // clamped index for the reads of a ragged tail
            const T *tb = terms + (((size_t)bc * nterms) << LOGN);
            T acc[NOUT][E];
            static_for<0, NOUT>([&](auto o) {
#pragma unroll
                for (int e = 0; e < E; ++e) acc[o.value][e] = 0;
            });
            T r[E];
            if constexpr (NEXT) FB::template gather<IO_RM>(r, tb, ebaseIO, false);
            for (uint32_t j = 0; j < nterms; ++j) {
                const uint32_t jn = j + 1 < nterms ? j + 1 : j;  // last iteration: harmless re-read
                T nx[NEXT ? E : 1];
                uint32_t tj = tid;
                if constexpr (OPAQUE) asm volatile("" : "+v"(tj));
                const uint32_t ebIO = !OPAQUE ? ebaseIO : pdep<CMIO>(tj), eb0 = !OPAQUE ? ebase0 : pdep<CM0>(tj);
                if constexpr (NEXT) FB::template gather<IO_RM>(nx, tb + ((size_t)jn << LOGN), ebIO, false);
                else FB::template gather<IO_RM>(r, tb + ((size_t)j << LOGN), ebIO, false);
#pragma unroll
                for (int e = 0; e < E; ++e) r[e] = Bfly<T, CLS>::load_fix(r[e], P);
                if constexpr (RM0 != IO_RM) {
                    FB::template scatter<IO_RM>(r, lds, ebIO, true);
                    F::wsync();
                    FB::template gather<RM0>(r, (const T *)lds, eb0, true);
                    F::wsync();
                }
                // NTT-domain values in layout RMM: canonical, or (CLS_FP) range-reduced doubles
                F::template pass<0, false, true, mul_fwd_fin<T, CLS>()>(r, lds, tj, twf, imgf, P);
                if constexpr (Bfly<T, CLS>::IS_FP) {
#pragma unroll
                    for (int e = 0; e < E; ++e) r[e] = Bfly<T, CLS>::reduce(r[e], P);
                }
                const uint32_t ebM = !OPAQUE ? ebaseM : pdep<CMM>(tj);
                // (the Montgomery products kept behind the transform: overlapped with its last stage the one-output shapes spill)
                if constexpr (sizeof(T) == 8 && mul_is_mont<CLS>()) __builtin_amdgcn_sched_barrier(0);
                static_for<0, NOUT>([&](auto o) {
                    mul_acc_key(acc[o.value], r, key_ntt + (((size_t)j * ostride + o.value) << LOGN), ebM, P);
                    if constexpr (Bfly<T, CLS>::IS_FP) {  // every product adds at most 0.875 p to the lazy accumulator
                        if ((j + 1u) % (uint32_t)Bfly<T, CLS>::ACC_REDUCE_EVERY == 0u) {
#pragma unroll
                            for (int e = 0; e < E; ++e) acc[o.value][e] = Bfly<T, CLS>::reduce(acc[o.value][e], P);
                        }
                    }
                });
                F::wsync();  // the forward transform's last exchange has been read before LDS is reused
                if constexpr (NEXT) {
#pragma unroll
                    for (int e = 0; e < E; ++e) r[e] = nx[e];
                }
            }
            static_for<0, NOUT>([&](auto o) {
                T(&a)[E] = acc[o.value];
                uint32_t to = tid;
                if constexpr (OPAQUE) asm volatile("" : "+v"(to));
                const uint32_t ebIO = !OPAQUE ? ebaseIO : pdep<CMIO>(to), ebL = !OPAQUE ? ebaseL : pdep<CML>(to);
#pragma unroll
                for (int e = 0; e < E; ++e) a[e] = chain_pre_inverse<T, CLS>(a[e], P);  // lazy accumulator -> inverse input
                I::template pass<0>(a, lds, to, twi, imgi, P);  // canonical coefficients, layout RML
                if constexpr (RML != IO_RM) {
                    F::wsync();
                    FB::template scatter<RML>(a, lds, ebL, true);
                    F::wsync();
                    FB::template gather<IO_RM>(a, (const T *)lds, ebIO, true);
                }
                T *dst = out + (((size_t)bc * ostride + o.value) << LOGN);
                if (accumulate) {
                    T old[E];
                    FB::template gather<IO_RM>(old, (const T *)dst, ebIO, false);
#pragma unroll
                    for (int e = 0; e < E; ++e) a[e] = add_mod<T>(old[e], a[e], P.p);
                }
                if (b < nb) FB::template scatter<IO_RM>(a, dst, ebIO, false);
                F::wsync();
            });
        }
    }
};

// The chain for the sizes without an LDS twiddle image (32-bit words, N = 8192 ... 32768): one batch element per workgroup,
// twiddles from the tables in global memory (L2), no persistent walk and no next-term prefetch (several workgroups per CU in
// different phases overlap loads and butterflies, as MulOne).  Same values as ExtWp.
template <class T, int LOGN, int CLS, int NOUT>
struct ExtOne {
    static constexpr int TPP = NttKernel<T, LOGN, false, CLS, false>::TPP;
    using X = ExtWp<T, LOGN, CLS, TPP, NOUT, plain_fam<T, LOGN, false>()>;
    using F = typename X::F;
    using I = typename X::I;
    using FB = typename X::FB;
    static constexpr int E = X::E;
    static constexpr uint32_t FULL = X::FULL, RM0 = X::RM0, RMM = X::RMM, RML = X::RML;

    static __device__ __forceinline__ void run(T *__restrict__ out, const T *__restrict__ terms, const T *__restrict__ key_ntt,
                                               const TwPair<T> *__restrict__ twf, const TwPair<T> *__restrict__ twi,
                                               const ModParams<T> &P, uint32_t nterms, bool accumulate, uint32_t ostride, T *lds) {
        constexpr uint32_t CM0 = FULL & ~RM0, CMM = FULL & ~RMM, CML = FULL & ~RML;
        const uint32_t tid = threadIdx.x, b = blockIdx.x;
        const T *tb = terms + (((size_t)b * nterms) << LOGN);
        T acc[NOUT][E];
        static_for<0, NOUT>([&](auto o) {
#pragma unroll
            for (int e = 0; e < E; ++e) acc[o.value][e] = 0;
        });
        for (uint32_t j = 0; j < nterms; ++j) {
            uint32_t tj = tid;   // per-term opaque copy: no offset of the transform stays live next to the accumulators
            asm volatile("" : "+v"(tj));
            T r[E];
            FB::template gather<RM0>(r, tb + ((size_t)j << LOGN), pdep<CM0>(tj), false);
#pragma unroll
            for (int e = 0; e < E; ++e) r[e] = Bfly<T, CLS>::load_fix(r[e], P);
            F::template pass<0, false, false, mul_fwd_fin<T, CLS>()>(r, lds, tj, twf, nullptr, P);
            if constexpr (Bfly<T, CLS>::IS_FP) {
#pragma unroll
                for (int e = 0; e < E; ++e) r[e] = Bfly<T, CLS>::reduce(r[e], P);
            }
            const uint32_t ebM = pdep<CMM>(tj);
            static_for<0, NOUT>([&](auto o) {
                X::mul_acc_key(acc[o.value], r, key_ntt + (((size_t)j * ostride + o.value) << LOGN), ebM, P);
                if constexpr (Bfly<T, CLS>::IS_FP) {
                    if ((j + 1u) % (uint32_t)Bfly<T, CLS>::ACC_REDUCE_EVERY == 0u) {
#pragma unroll
                        for (int e = 0; e < E; ++e) acc[o.value][e] = Bfly<T, CLS>::reduce(acc[o.value][e], P);
                    }
                }
            });
            F::wsync();  // the forward transform's last exchange has been read before LDS is reused
        }
        static_for<0, NOUT>([&](auto o) {
            T(&a)[E] = acc[o.value];
            uint32_t to = tid;
            asm volatile("" : "+v"(to));
#pragma unroll
            for (int e = 0; e < E; ++e) a[e] = chain_pre_inverse<T, CLS>(a[e], P);
            I::template pass<0, false, false>(a, lds, to, twi, nullptr, P);  // canonical coefficients, layout RML
            T *dst = out + (((size_t)b * ostride + o.value) << LOGN);
            if (accumulate) {
                T old[E];
                FB::template gather<RML>(old, (const T *)dst, pdep<CML>(to), false);
#pragma unroll
                for (int e = 0; e < E; ++e) a[e] = add_mod<T>(old[e], a[e], P.p);
            }
            FB::template scatter<RML>(a, dst, pdep<CML>(to), false);
            F::wsync();
        });
    }
};

template <class T, int LOGN, int CLS, int NOUT>
__global__ __launch_bounds__((ExtOne<T, LOGN, CLS, NOUT>::TPP)) void ext_kernel_one(T *__restrict__ out, const T *__restrict__ terms,
                                                                                 const T *__restrict__ key_ntt,
                                                                                 const TwPair<T> *__restrict__ twf,
                                                                                 const TwPair<T> *__restrict__ twi,
                                                                                 const ModParams<T> P, uint32_t nterms,
                                                                                 uint32_t accumulate, uint32_t ostride) {
    using K = ExtOne<T, LOGN, CLS, NOUT>;
    __shared__ __attribute__((aligned(16))) T lds[K::FB::LDS_WORDS_1];
    K::run(out, terms, key_ntt, twf, twi, P, nterms, accumulate != 0, ostride, lds);
}

template <class T, int LOGN, int CLS, int WPB, int WPW, int NOUT, int FAM = 0>
__global__ __launch_bounds__(WPB, WPW) void ext_kernel_wp(T *__restrict__ out, const T *__restrict__ terms,
                                                      const T *__restrict__ key_ntt, const TwPair<T> *__restrict__ twf,
                                                      const TwPair<T> *__restrict__ twi, const ModParams<T> P,
                                                      uint32_t nb, uint32_t nterms, uint32_t accumulate, uint32_t ostride) {
    using K = ExtWp<T, LOGN, CLS, WPB, NOUT, FAM>;
    __shared__ __attribute__((aligned(16))) T lds[(size_t)K::PPB * K::FB::LDS_WORDS_1];
    __shared__ __attribute__((aligned(16))) TwPair<T> imgf[K::FB::IMG_ENTRIES];
    __shared__ __attribute__((aligned(16))) TwPair<T> imgi[K::IB::IMG_ENTRIES];
    K::run(out, terms, key_ntt, twf, twi, P, nb, nterms, accumulate != 0, ostride, lds, imgf, imgi);
}

template <class T, int LOGN, bool INV, int CLS, bool SUB, int FAM = 0>
__global__ __launch_bounds__((NttKernel<T, LOGN, INV, CLS, SUB, FAM>::BLOCK)) void ntt_kernel(
    T *__restrict__ data, const TwPair<T> *__restrict__ tw, const ModParams<T> P, uint32_t nsub, uint32_t depth) {
    using K = NttKernel<T, LOGN, INV, CLS, SUB, FAM>;
    __shared__ __attribute__((aligned(16))) T lds[K::LDS_ELEMS];
    K::run(data, tw, P, nsub, depth, lds);
}

}  // namespace cntt
