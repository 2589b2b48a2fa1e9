// "Wave block" transforms for 64-bit words, N = 4096 ... 16384 (schedule family 2 of gen_sched.py).
//
// Replaces the same reference engines as ntt_kernel.hpp (fwd: src/prime64/shoup.rs:544-706, inv: :1306-1468) with the
// same stage / twiddle indexing; what changes is who owns what.  A polynomial of N = 2^LOGN words belongs to one
// workgroup of N/16 threads and crosses LDS between wavefronts exactly ONCE:
//   forward   pass 0: the top LOGN-10 stages on registers loaded straight from HBM at a stride of N/16 words
//                     (consecutive lanes on consecutive words; every twiddle of these stages is workgroup-uniform);
//             LDS transpose -- the one exchange between wavefronts (workgroup barrier);
//             passes 1-3: every wavefront owns a contiguous 1024-word block and runs the N = 1024 schedule on it
//                     (3 + 4 + 3 stages, exchanges through the wavefront's own 8 KiB of the buffer: no barrier; the
//                     first of them has wave-uniform twiddles: scalar loads), then stores its block coalesced.
//   inverse   the mirror image: block passes first, one transpose, the top stages last.
// A second barrier per polynomial only keeps the next polynomial's first LDS write behind everybody's last read of the
// current one; a whole pass of butterflies lies between the two, so nobody waits at it for long.
// (The round-2 walk of these sizes -- NttWp with one polynomial per workgroup, removed in round 3 -- crossed 9-11
// workgroup barriers per polynomial with three cross-wave exchanges and two cross-wave I/O transposes, and its compiler-
// placed vmcnt waits retired every prefetch right after issue: profiles/r03_blk_vs_round2.txt.)
#pragma once
#include "ntt_kernel.hpp"

// (The timing-only ablation switches and phase stamps of tools/blk_lab.hip are NOT in this header: tools/blk_lab.sh applies
// tools/blk_lab.patch to a scratch copy.)

namespace cntt {

// SUB: the transform is one half of a 2^(LOGN+1)-point polynomial (Ntt32k below): table offsets as NttKernel's SUB mode
// with depth 1 and the workgroup-uniform prefix `qpre` = half << LOGN.
template <class T, int LOGN, bool INV, int CLS, int TWC = 2, bool SUB = false>
struct NttBlk {
    using B = NttKernel<T, LOGN, INV, CLS, SUB, 2>;
    using S = typename B::S;
    // 64-bit words: 1024-word blocks, 16 coefficients per thread; 32-bit words (round 4): 2048-word blocks, 32 per thread
    static_assert(B::NPASS == 4 && B::LOGE == (sizeof(T) == 8 ? 4 : 5), "wave-block schedules: four passes, 8 KiB per wavefront");
    static constexpr int E = B::E, TPP = B::TPP, WPB = B::TPP;
    static constexpr uint32_t FULL = B::FULL;
    static constexpr uint32_t BLKMASK = sizeof(T) == 8 ? 0x3ffu : 0x7ffu;   // index bits inside a wavefront's block
    static constexpr uint32_t BLK_IO = sizeof(T) == 8 ? 0x381u : 0x703u;   // a wavefront's block, 16 bytes per lane on consecutive addresses
    static constexpr uint32_t RM0 = S::RMASK[0], RM1 = S::RMASK[1], RM2 = S::RMASK[2], RM3 = S::RMASK[3];
    static constexpr uint32_t LOAD_RM = INV ? BLK_IO : RM0;   // HBM layouts: the top pass reads / writes its own layout
    static constexpr uint32_t STORE_RM = INV ? RM3 : BLK_IO;
    static constexpr int HARD = INV ? 2 : 0;                  // the exchange after this pass crosses wavefronts
    static_assert(((INV ? RM0 : RM3) & ~BLKMASK) == 0 && ((INV ? RM3 : RM0) & ~BLKMASK) == (FULL & ~BLKMASK), "block passes / top pass");

    template <uint32_t RM> static constexpr int nv() { return B::template vec_elems<RM>(); }

    template <bool PRIV> static __device__ __forceinline__ void xsync() {
        if constexpr (PRIV) {  // LDS operations of one wavefront execute in order: only the compiler must not reorder
            asm volatile("" ::: "memory");
            __builtin_amdgcn_wave_barrier();
            asm volatile("" ::: "memory");
        } else {  // raw barrier behind an LDS-only wait: __syncthreads() would drain the prefetched global loads too
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
        }
    }

    // ---- asynchronous prefetch of the next polynomial (see NttKernel::gather_async): 16- or 8-byte loads ----
    using Vec4 = __attribute__((ext_vector_type(4))) uint32_t;
    using Vec2 = __attribute__((ext_vector_type(2))) uint32_t;
    template <uint32_t RM> struct Pf {
        static constexpr int NV = nv<RM>(), NVEC = E / NV;
        static constexpr bool B16 = NV * sizeof(T) == 16;   // 16-byte loads (8-byte ones otherwise)
        static_assert(NV * sizeof(T) == 16 || NV * sizeof(T) == 8, "prefetch vectors of 8 or 16 bytes");
        using V = typename std::conditional<B16, Vec4, Vec2>::type;
        template <int JV = 0, bool NT = false> static __device__ __forceinline__ void issue(V (&v)[NVEC], const T *tile, uint32_t voff) {
            if constexpr (JV < NVEC) {
                constexpr uint32_t BYTE = cdep((uint32_t)(JV * NV), RM) * (uint32_t)sizeof(T);
                constexpr uint32_t WIN = BYTE & ~4095u, IMM = BYTE & 4095u;
                const char *base = reinterpret_cast<const char *>(tile) + WIN;
                if constexpr (B16 && NT)
                    asm volatile("global_load_dwordx4 %0, %1, %2 offset:%3 nt" : "=v"(v[JV]) : "v"(voff), "s"(base), "n"(IMM) : "memory");
                else if constexpr (B16)
                    asm volatile("global_load_dwordx4 %0, %1, %2 offset:%3" : "=v"(v[JV]) : "v"(voff), "s"(base), "n"(IMM) : "memory");
                else if constexpr (NT)
                    asm volatile("global_load_dwordx2 %0, %1, %2 offset:%3 nt" : "=v"(v[JV]) : "v"(voff), "s"(base), "n"(IMM) : "memory");
                else
                    asm volatile("global_load_dwordx2 %0, %1, %2 offset:%3" : "=v"(v[JV]) : "v"(voff), "s"(base), "n"(IMM) : "memory");
                issue<JV + 1, NT>(v, tile, voff);
            }
        }
        // policy chosen at run time (ModParams::stream; NttKernel::gather_async_rt)
        static __device__ __forceinline__ void issue_rt(V (&v)[NVEC], const T *tile, uint32_t voff, bool nt) {
            if (nt) issue<0, true>(v, tile, voff);
            else issue<0, false>(v, tile, voff);
        }
        // every destination vector is a tied operand of the ONE waiting statement
        template <int YOUNGER> static __device__ __forceinline__ void wait(V (&v)[NVEC]) {
            if constexpr (NVEC == 8) {
                asm volatile("s_waitcnt vmcnt(%8)"
                             : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7])
                             : "n"(YOUNGER)
                             : "memory");
            } else {
                static_assert(NVEC == 16, "16 coefficients per thread");
                asm volatile("s_waitcnt vmcnt(%16)"
                             : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7]),
                               "+v"(v[8]), "+v"(v[9]), "+v"(v[10]), "+v"(v[11]), "+v"(v[12]), "+v"(v[13]), "+v"(v[14]), "+v"(v[15])
                             : "n"(YOUNGER)
                             : "memory");
            }
        }
        static __device__ __forceinline__ void unpack(T (&r)[E], const V (&v)[NVEC]) {
#pragma unroll
            for (int j = 0; j < E; j += NV) {
                if constexpr (sizeof(T) == 8) {
                    r[j] = (T)v[j / NV][0] | ((T)v[j / NV][1] << 32);
                    if constexpr (NV == 2) r[j + 1] = (T)v[j / NV][2] | ((T)v[j / NV][3] << 32);
                } else {
                    // explicit moves AFTER the wait (NttKernel::unpack_async: a plain copy lets hipcc satisfy the wait statement's tied
                    // operand by copying the not-yet-landed destination register before the s_waitcnt)
#pragma unroll
                    for (int i = 0; i < NV; ++i) {
                        uint32_t x;
                        asm("v_mov_b32 %0, %1" : "=v"(x) : "v"(v[j / NV][i]));
                        r[j + i] = (T)x;
                    }
                }
            }
        }
    };

    template <int K, bool NORM>
    static __device__ __forceinline__ void stages(T (&r)[E], uint32_t tidv, const TwPair<T> *__restrict__ tw, const ModParams<T> &P,
                                                  uint32_t qpre = 0u) {
        constexpr uint32_t CM = FULL & ~S::RMASK[K];
        B::template stages<K, 0, false, NORM, TWC>(r, pdep<CM>(tidv), qpre, SUB ? 1u : 0u, tw, P, tidv, nullptr);
    }
    // exchange between pass K and pass K + 1
    template <int K, bool PRE_PRIV> static __device__ __forceinline__ void exch(T (&r)[E], T *lds, uint32_t tidv) {
        constexpr uint32_t RA = S::RMASK[K], RB = S::RMASK[K + 1];
        xsync<PRE_PRIV>();  // what the buffer held has been read
        B::template scatter<RA>(r, lds, pdep<FULL & ~RA>(tidv), true);
        xsync<K != HARD>();
        B::template gather<RB>(r, (const T *)lds, pdep<FULL & ~RB>(tidv), true);
    }

    // NORM: the last inverse stage also applies 1/N (fused products).
    // HOOK: called where the scalar-twiddle part of the transform begins -- at the start of the forward transform, before
    // the third pass of the inverse.  That is where a kernel issues its asynchronous prefetch: vmcnt retires in order, so
    // the first wait for a thread-dependent twiddle load behind the prefetch would also wait for the prefetch; issued
    // here it stays in flight across two passes of butterflies (and is live in registers only that long).
    struct NoHook {
        __device__ __forceinline__ void operator()() const {}
    };
    template <bool NORM = false, class HOOK = NoHook>
    static __device__ __forceinline__ void transform(T (&r)[E], T *lds, uint32_t tidv, const TwPair<T> *__restrict__ tw,
                                                     const ModParams<T> &P, const HOOK &hook = HOOK{}, uint32_t qpre = 0u) {
        if constexpr (!INV) {
            hook();
            stages<0, false>(r, tidv, tw, P, qpre);
            exch<0, false>(r, lds, tidv);  // behind everybody's last read of the previous polynomial; then the one barrier
            stages<1, false>(r, tidv, tw, P, qpre);
            exch<1, true>(r, lds, tidv);
            stages<2, false>(r, tidv, tw, P, qpre);
            exch<2, true>(r, lds, tidv);
            stages<3, false>(r, tidv, tw, P, qpre);
        } else {
            stages<0, false>(r, tidv, tw, P, qpre);
            exch<0, true>(r, lds, tidv);
            stages<1, false>(r, tidv, tw, P, qpre);
            exch<1, true>(r, lds, tidv);
            hook();
            stages<2, false>(r, tidv, tw, P, qpre);
            exch<2, true>(r, lds, tidv);   // the one barrier sits between its scatter and its gather
            stages<3, NORM>(r, tidv, tw, P, qpre);
        }
    }

    // PREFETCH = false: every polynomial is read with ordinary loads where its transform begins -- for the shapes whose register
    // need makes hipcc spill (32-bit words on doubles, the strict class's inverse at 32 coefficients per thread): harmless next to
    // ordinary loads, fatal next to asynchronous ones (tests/test_async_load_guard.py)
    template <bool PREFETCH = true>
    static __device__ __forceinline__ void run(T *__restrict__ data, const TwPair<T> *__restrict__ tw, const ModParams<T> &P,
                                               uint32_t nsub, T *lds) {
        using PF = Pf<LOAD_RM>;
        constexpr uint32_t CML = FULL & ~LOAD_RM, CMS = FULL & ~STORE_RM;
        constexpr int NST = E / nv<STORE_RM>();  // store instructions per polynomial (younger than the prefetch)
        const uint32_t tid = threadIdx.x;
        const bool nt = P.stream != 0;   // the batch streams: non-temporal tile loads and stores (NttKernel::scatter_tile)
        T r[E];
#pragma unroll
        for (int j = 0; j < E; ++j) r[j] = 0;
        uint32_t tile = blockIdx.x;
        // The first polynomial takes the same asynchronous path as every later one.  A plain C++ load here would leave
        // hipcc's vmcnt model with loads "in flight" at the loop header: it then waits for them INSIDE the loop, with
        // counts that on every later iteration drain the prefetch of the next polynomial right after it was issued
        // (no overlap of HBM and butterflies at all: measured 102 instead of 80 ns per polynomial at N = 16384).
        if constexpr (PREFETCH) {
            if (tile < nsub) {
                typename PF::V v0[PF::NVEC];
                PF::issue_rt(v0, (const T *)(data + ((size_t)tile << LOGN)), pdep<CML>(tid) * (uint32_t)sizeof(T), nt);
                PF::template wait<0>(v0);
                PF::unpack(r, v0);
            }
        }
        for (; tile < nsub; tile += gridDim.x) {
            // every address of the body is recomputed from an opaque copy of the thread index: left alone hipcc hoists
            // them all out of the loop and spills (128 VGPRs: sixteen wavefronts per CU)
            uint32_t tidv = tid;
            asm volatile("" : "+v"(tidv));
            const uint32_t tnext = tile + gridDim.x;
            const bool more = tnext < nsub;  // workgroup-uniform
            T *tbase = data + ((size_t)tile << LOGN);
            typename PF::V vn[PF::NVEC];
            auto prefetch = [&]() {
                if constexpr (PREFETCH) {
                    if (more) PF::issue_rt(vn, (const T *)(data + ((size_t)tnext << LOGN)), pdep<CML>(tidv) * (uint32_t)sizeof(T), nt);
                }
            };
            if constexpr (!PREFETCH) B::template gather_tile<LOAD_RM>(r, (const T *)tbase, pdep<CML>(tidv) * (uint32_t)sizeof(T));
#pragma unroll
            for (int j = 0; j < E; ++j) r[j] = Bfly<T, CLS>::load_fix(r[j], P);
            if constexpr (!INV) {
                transform<false>(r, lds, tidv, tw, P, prefetch);
#pragma unroll
                for (int j = 0; j < E; ++j) r[j] = Bfly<T, CLS>::finish_fwd(r[j], P);
                if constexpr (STORE_RM != RM3) {
                    xsync<true>();  // block transpose into the coalesced layout (wave-private)
                    B::template scatter<RM3>(r, lds, pdep<FULL & ~RM3>(tidv), true);
                    xsync<true>();
                    B::template gather<BLK_IO>(r, (const T *)lds, pdep<CMS>(tidv), true);
                }
            } else {
                xsync<false>();  // everybody has read the previous polynomial's transpose
                if constexpr (LOAD_RM != RM0) {
                    B::template scatter<BLK_IO>(r, lds, pdep<CML>(tidv), true);
                    xsync<true>();
                    B::template gather<RM0>(r, (const T *)lds, pdep<FULL & ~RM0>(tidv), true);
                }
                transform<false>(r, lds, tidv, tw, P, prefetch);
#pragma unroll
                for (int j = 0; j < E; ++j) r[j] = Bfly<T, CLS>::finish_inv(r[j], P);
            }
            B::template scatter_tile_rt<STORE_RM>(r, tbase, pdep<CMS>(tidv) * (uint32_t)sizeof(T), nt);
            if constexpr (PREFETCH) {
                if (more) {
                    PF::template wait<NST>(vn);
                    PF::unpack(r, vn);
                }
            }
        }
    }
};

template <class T, int LOGN, bool INV, int CLS, int WPW, int TWC = 2, bool PREFETCH = true>
__global__ __launch_bounds__((NttBlk<T, LOGN, INV, CLS>::WPB), WPW) void ntt_kernel_blk(T *__restrict__ data,
                                                                                     const TwPair<T> *__restrict__ tw,
                                                                                     const ModParams<T> P, uint32_t nsub) {
    using K = NttBlk<T, LOGN, INV, CLS, TWC>;
    __shared__ __attribute__((aligned(16))) T lds[K::B::LDS_WORDS_1];
    K::template run<PREFETCH>(data, tw, P, nsub, lds);
}

// -------------------------------------------------------------------------------------------------
// N = 32768 (64-bit words) in ONE pass over HBM.  The 256 KiB polynomial does not fit LDS, but it fits the registers of a
// 1024-thread workgroup (32 coefficients per thread): the stage on index bit 14 pairs coefficient e with e + 16384 and
// needs no exchange when a thread holds both; after it the two halves are independent 16384-point transforms with table
// offsets of their own (SUB mode, depth 1), which go through the wave-block walk above one after the other, the second
// half parked in registers while the first one owns the 128 KiB exchange buffer.  Inverse: the mirror image.
// Replaces global_stage_kernel + two LDS-resident sub-transforms (two passes over HBM): src/prime64/shoup.rs:660-682 is the
// reference's own depth-first split with the same stage / twiddle indexing.  No register prefetch (the parked half takes
// its place): a polynomial's loads are exposed once per ~50 us of butterflies.
// -------------------------------------------------------------------------------------------------
template <class T, bool INV, int CLS, int TWC = 1>
struct Ntt32k {
    static constexpr int LOGH = 14;
    using H = NttBlk<T, LOGH, INV, CLS, TWC, true>;
    using HB = typename H::B;
    static constexpr int E = H::E, WPB = H::WPB;
    static constexpr uint32_t FULL = H::FULL, TOP = INV ? H::RM3 : H::RM0;   // the top pass's layout: HBM side of a half
    static constexpr uint32_t BLK_IO = H::BLK_IO;

    static __device__ __forceinline__ void run(T *__restrict__ data, const TwPair<T> *__restrict__ tw, const ModParams<T> &P,
                                               uint32_t nsub, T *lds) {
        constexpr uint32_t CMT = FULL & ~TOP, CMB = FULL & ~BLK_IO;
        const uint32_t tid = threadIdx.x;
        const TwPair<T> w14 = tw[1];   // the stage on bit 14: one twiddle for all of its butterflies (table entry 2^0 + 0)
        for (uint32_t tile = blockIdx.x; tile < nsub; tile += gridDim.x) {
            uint32_t tidv = tid;
            asm volatile("" : "+v"(tidv));
            T *base = data + ((size_t)tile << (LOGH + 1));
            T a[E], b[E];
            if constexpr (!INV) {
                HB::template gather_tile<TOP>(a, (const T *)base, pdep<CMT>(tidv) * (uint32_t)sizeof(T));
                HB::template gather_tile<TOP>(b, (const T *)(base + ((size_t)1 << LOGH)), pdep<CMT>(tidv) * (uint32_t)sizeof(T));
#pragma unroll
                for (int j = 0; j < E; ++j) {
                    a[j] = Bfly<T, CLS>::load_fix(a[j], P);
                    b[j] = Bfly<T, CLS>::load_fix(b[j], P);
                    Bfly<T, CLS>::template fwd<true, true>(a[j], b[j], w14.w, w14.ws, P);   // the transform's first stage
                    if constexpr (Bfly<T, CLS>::IS_FP) {  // the halves' reduction schedules assume inputs no larger than canonical ones
                        a[j] = Bfly<T, CLS>::reduce(a[j], P);
                        b[j] = Bfly<T, CLS>::reduce(b[j], P);
                    }
                }
                // (the second half recomputes its addresses from a fresh opaque copy of the thread index: shared with the
                // first half's they would all stay live across it, next to the parked coefficients)
                uint32_t t2 = tidv;
                asm volatile("" : "+v"(t2));
                half_fwd(a, lds, tidv, tw, P, 0u, base);
                half_fwd(b, lds, t2, tw, P, 1u << LOGH, base + ((size_t)1 << LOGH));
            } else {
                uint32_t t2 = tidv;
                asm volatile("" : "+v"(t2));
                half_inv(a, lds, tidv, tw, P, 0u, base);
                half_inv(b, lds, t2, tw, P, 1u << LOGH, base + ((size_t)1 << LOGH));
#pragma unroll
                for (int j = 0; j < E; ++j) {
                    if constexpr (Bfly<T, CLS>::IS_FP) {  // whatever the halves' schedules left: back to |v| <= p/2 before the sum
                        a[j] = Bfly<T, CLS>::reduce(a[j], P);
                        b[j] = Bfly<T, CLS>::reduce(b[j], P);
                    }
                    Bfly<T, CLS>::template inv<true>(a[j], b[j], w14.w, w14.ws, P);
                    a[j] = Bfly<T, CLS>::finish_inv(a[j], P);
                    b[j] = Bfly<T, CLS>::finish_inv(b[j], P);
                }
                HB::template scatter_tile<TOP>(a, base, pdep<CMT>(tidv) * (uint32_t)sizeof(T));
                HB::template scatter_tile<TOP>(b, base + ((size_t)1 << LOGH), pdep<CMT>(tidv) * (uint32_t)sizeof(T));
            }
        }
    }
    // forward: a half (already past the stage on bit 14) through the wave-block walk, canonicalised and stored
    static __device__ __forceinline__ void half_fwd(T (&r)[E], T *lds, uint32_t tidv, const TwPair<T> *__restrict__ tw,
                                                    const ModParams<T> &P, uint32_t qpre, T *hbase) {
        H::template transform<false>(r, lds, tidv, tw, P, typename H::NoHook{}, qpre);
#pragma unroll
        for (int j = 0; j < E; ++j) r[j] = Bfly<T, CLS>::finish_fwd(r[j], P);
        H::template xsync<true>();  // block transpose into the coalesced layout (wave-private)
        HB::template scatter<H::RM3>(r, lds, pdep<FULL & ~H::RM3>(tidv), true);
        H::template xsync<true>();
        HB::template gather<BLK_IO>(r, (const T *)lds, pdep<FULL & ~BLK_IO>(tidv), true);
        HB::template scatter_tile<BLK_IO>(r, hbase, pdep<FULL & ~BLK_IO>(tidv) * (uint32_t)sizeof(T));
    }
    // inverse: a half loaded block-wise, through the walk up to (not including) the stage on bit 14; values stay lazy
    static __device__ __forceinline__ void half_inv(T (&r)[E], T *lds, uint32_t tidv, const TwPair<T> *__restrict__ tw,
                                                    const ModParams<T> &P, uint32_t qpre, const T *hbase) {
        HB::template gather_tile<BLK_IO>(r, hbase, pdep<FULL & ~BLK_IO>(tidv) * (uint32_t)sizeof(T));
#pragma unroll
        for (int j = 0; j < E; ++j) r[j] = Bfly<T, CLS>::load_fix(r[j], P);
        H::template xsync<false>();  // everybody has read what the buffer held (the other half's transpose)
        HB::template scatter<BLK_IO>(r, lds, pdep<FULL & ~BLK_IO>(tidv), true);
        H::template xsync<true>();
        HB::template gather<H::RM0>(r, (const T *)lds, pdep<FULL & ~H::RM0>(tidv), true);
        H::template transform<false>(r, lds, tidv, tw, P, typename H::NoHook{}, qpre);
    }
};

template <class T, bool INV, int CLS, int WPW>
__global__ __launch_bounds__((Ntt32k<T, INV, CLS>::WPB), WPW) void ntt_kernel_32k(T *__restrict__ data, const TwPair<T> *__restrict__ tw,
                                                                                const ModParams<T> P, uint32_t nsub) {
    using K = Ntt32k<T, INV, CLS>;
    __shared__ __attribute__((aligned(16))) T lds[K::HB::LDS_WORDS_1];
    K::run(data, tw, P, nsub, lds);
}

// -------------------------------------------------------------------------------------------------
// Fused negacyclic product against a pre-transformed operand on the wave-block walk (the large-N counterpart of MulWp):
//     lhs <- inv( mul_assign_normalize( fwd(lhs), rhs_ntt ) )        src/prime64.rs:794, :947-1033, :872
// The forward transform ends and the inverse transform starts inside a wavefront's own 1024-word block (mirror
// schedules: the forward's last register layout is the inverse's first), so the pointwise product happens in registers
// between them and a polynomial crosses THREE workgroup barriers for two transforms; 1/N rides in the last inverse
// stage (Bfly::inv_norm).  3 N words of HBM traffic instead of 7 N.
// -------------------------------------------------------------------------------------------------
// PREFETCH: the next polynomial's lhs is loaded asynchronously into registers during the inverse transform's last two
// passes (a kernel whose shape cannot hold it without spilling must run without: an asynchronous load must never meet a
// spilled register -- the CPU tests read the code objects).
template <class T, int LOGN, int CLS, int TWC = 2, bool PREFETCH = true>
struct MulBlk {
    using F = NttBlk<T, LOGN, false, CLS, TWC>;
    using I = NttBlk<T, LOGN, true, CLS, TWC>;
    using FB = typename F::B;
    static constexpr int E = F::E, WPB = F::WPB;
    static constexpr uint32_t FULL = F::FULL;
    static constexpr uint32_t RMIO = F::RM0;                    // HBM layout of lhs: the top pass's own (fwd loads, inv stores)
    static constexpr uint32_t RMM = F::RM3;                     // NTT-domain layout (inside a wavefront's block)
    static_assert(RMM == I::RM0 && RMIO == I::RM3, "forward and inverse schedules must mirror each other");

    static __device__ __forceinline__ void run(T *__restrict__ lhs, const T *__restrict__ rhs_ntt,
                                               const TwPair<T> *__restrict__ twf, const TwPair<T> *__restrict__ twi,
                                               const ModParams<T> &P, uint32_t nsub, T *lds) {
        using PF = typename F::template Pf<RMIO>;
        constexpr uint32_t CMIO = FULL & ~RMIO, CMM = FULL & ~RMM;
        constexpr int NST = E / F::template nv<RMIO>();
        const uint32_t tid = threadIdx.x;
        T r[E];
#pragma unroll
        for (int j = 0; j < E; ++j) r[j] = 0;
        uint32_t tile = blockIdx.x;
        if constexpr (PREFETCH) {
            if (tile < nsub) {  // asynchronous like every later polynomial (see NttBlk::run)
                typename PF::V v0[PF::NVEC];
                PF::issue(v0, (const T *)(lhs + ((size_t)tile << LOGN)), pdep<CMIO>(tid) * (uint32_t)sizeof(T));
                PF::template wait<0>(v0);
                PF::unpack(r, v0);
            }
        }
        for (; tile < nsub; tile += gridDim.x) {
            uint32_t tidv = tid;
            asm volatile("" : "+v"(tidv));
            const uint32_t tnext = tile + gridDim.x;
            const bool more = tnext < nsub;
            T *tbase = lhs + ((size_t)tile << LOGN);
            typename PF::V vn[PF::NVEC];
            if constexpr (!PREFETCH) FB::template gather_tile<RMIO>(r, (const T *)tbase, pdep<CMIO>(tidv) * (uint32_t)sizeof(T));
            // issued where the inverse transform's scalar-twiddle passes begin (NttBlk::transform): in flight -- and live
            // in registers -- across its last two passes only, clear of the products and of every thread-dependent twiddle
            auto prefetch = [&]() {
                if constexpr (PREFETCH) {
                    if (more) PF::issue(vn, (const T *)(lhs + ((size_t)tnext << LOGN)), pdep<CMIO>(tidv) * (uint32_t)sizeof(T));
                }
            };
#pragma unroll
            for (int j = 0; j < E; ++j) r[j] = Bfly<T, CLS>::load_fix(r[j], P);
            F::transform(r, lds, tidv, twf, P);
            if constexpr (mul_fwd_fin<T, CLS>()) {   // (lazy class: the Montgomery product below takes the lazy outputs, ntt_arith.hpp mul_fused)
#pragma unroll
                for (int j = 0; j < E; ++j) r[j] = Bfly<T, CLS>::finish_fwd(r[j], P);
            }
            {
                // rhs_ntt is taken four coefficients at a time, the loads of the next four in flight behind the products of
                // the current ones, with the phases pinned (asm statements on r, compiler-level fences around the loads):
                // left alone hipcc overlaps the forward transform's tail, all sixteen products and the inverse transform's
                // head, needs 160 VGPRs, and at the 128 of this shape spills the destinations of the prefetch in flight.
                constexpr int CH = 4, NCH = E / CH;
                const T *rt = rhs_ntt + ((size_t)tile << LOGN);
                const uint32_t roff = pdep<CMM>(tidv) * (uint32_t)sizeof(T);
#pragma unroll
                for (int j = 0; j < E; ++j) asm volatile("" : "+v"(r[j]));
                T bc[2][CH];
                asm volatile("" ::: "memory");
                FB::template gather_tile_part<RMM, 0, CH>(bc[0], rt, roff);
                static_for<0, NCH>([&](auto c) {
                    constexpr int C = decltype(c)::value;
                    asm volatile("" ::: "memory");
                    if constexpr (C + 1 < NCH) FB::template gather_tile_part<RMM, (C + 1) * CH, CH>(bc[(C + 1) & 1], rt, roff);
                    asm volatile("" ::: "memory");
#pragma unroll
                    for (int i = 0; i < CH; ++i) {
                        r[C * CH + i] = mul_fused<T, CLS>(r[C * CH + i], bc[C & 1][i], P);  // 1/N: inside the last inverse stage
                        asm volatile("" : "+v"(r[C * CH + i]));
                    }
                });
            }
            // the inverse starts in the wavefront's own block: nothing of another wavefront is touched until its transpose
            uint32_t ti = tidv;   // fresh opaque copy: no address of the forward half stays live into the inverse half
            asm volatile("" : "+v"(ti));
            I::template transform<true>(r, lds, ti, twi, mul_inv_params<T, CLS>(P), prefetch);
#pragma unroll
            for (int j = 0; j < E; ++j) r[j] = Bfly<T, CLS>::finish_inv(r[j], P);
            FB::template scatter_tile<RMIO>(r, tbase, pdep<CMIO>(tidv) * (uint32_t)sizeof(T));
            if constexpr (PREFETCH) {
                if (more) {
                    PF::template wait<NST>(vn);
                    PF::unpack(r, vn);
                }
            }
        }
    }
};

template <class T, int LOGN, int CLS, int WPW, int TWC = 2, bool PREFETCH = true>
__global__ __launch_bounds__((MulBlk<T, LOGN, CLS>::WPB), WPW) void mul_kernel_blk(T *__restrict__ lhs, const T *__restrict__ rhs_ntt,
                                                                                const TwPair<T> *__restrict__ twf,
                                                                                const TwPair<T> *__restrict__ twi,
                                                                                const ModParams<T> P, uint32_t nsub) {
    using K = MulBlk<T, LOGN, CLS, TWC, PREFETCH>;
    __shared__ __attribute__((aligned(16))) T lds[K::FB::LDS_WORDS_1];
    K::run(lhs, rhs_ntt, twf, twi, P, nsub, lds);
}

// -------------------------------------------------------------------------------------------------
// The fused product at N = 32768 (Ntt32k's counterpart of MulBlk): one pass over HBM for
//     lhs <- inv( mul_assign_normalize( fwd(lhs), rhs_ntt ) ).
// The stage on index bit 14 in registers, then each 16384-point half in turn -- forward walk, products against the matching
// half of rhs_ntt in the wavefronts' block layout, inverse walk -- while the other half is parked in registers; the inverse
// stage on bit 14 is the transform's last one and carries 1/N (Bfly::inv_norm).  3 N words of HBM traffic instead of 7 N.
// Ordinary loads (no asynchronous prefetch: the parked half takes its place).
// -------------------------------------------------------------------------------------------------
template <class T, int CLS, int TWC = 1>
struct Mul32k {
    static constexpr int LOGH = 14;
    using HF = NttBlk<T, LOGH, false, CLS, TWC, true>;
    using HI = NttBlk<T, LOGH, true, CLS, TWC, true>;
    using HB = typename HF::B;
    static constexpr int E = HF::E, WPB = HF::WPB;
    static constexpr uint32_t FULL = HF::FULL, TOP = HF::RM0, RMM = HF::RM3;
    static_assert(RMM == HI::RM0 && TOP == HI::RM3, "forward and inverse schedules must mirror each other");

    // one half, past the forward stage on bit 14: forward walk, products, inverse walk up to (not including) that stage
    static __device__ __forceinline__ void half(T (&r)[E], T *lds, uint32_t tidv, const T *__restrict__ rhs_half,
                                                const TwPair<T> *__restrict__ twf, const TwPair<T> *__restrict__ twi,
                                                const ModParams<T> &P, uint32_t qpre) {
        HF::template transform<false>(r, lds, tidv, twf, P, typename HF::NoHook{}, qpre);
        if constexpr (mul_fwd_fin<T, CLS>()) {
#pragma unroll
            for (int j = 0; j < E; ++j) r[j] = Bfly<T, CLS>::finish_fwd(r[j], P);
        }
        {   // rhs_ntt four coefficients at a time, phases pinned (see MulBlk)
            constexpr int CH = 4, NCH = E / CH;
            const uint32_t roff = pdep<FULL & ~RMM>(tidv) * (uint32_t)sizeof(T);
#pragma unroll
            for (int j = 0; j < E; ++j) asm volatile("" : "+v"(r[j]));
            T bc[2][CH];
            asm volatile("" ::: "memory");
            HB::template gather_tile_part<RMM, 0, CH>(bc[0], rhs_half, roff);
            static_for<0, NCH>([&](auto c) {
                constexpr int C = decltype(c)::value;
                asm volatile("" ::: "memory");
                if constexpr (C + 1 < NCH) HB::template gather_tile_part<RMM, (C + 1) * CH, CH>(bc[(C + 1) & 1], rhs_half, roff);
                asm volatile("" ::: "memory");
#pragma unroll
                for (int i = 0; i < CH; ++i) {
                    r[C * CH + i] = mul_fused<T, CLS>(r[C * CH + i], bc[C & 1][i], P);
                    asm volatile("" : "+v"(r[C * CH + i]));
                }
            });
        }
        uint32_t ti = tidv;   // fresh opaque copy: no address of the forward walk stays live into the inverse walk
        asm volatile("" : "+v"(ti));
        HI::template transform<false>(r, lds, ti, twi, P, typename HI::NoHook{}, qpre);
    }

    static __device__ __forceinline__ void run(T *__restrict__ lhs, const T *__restrict__ rhs_ntt,
                                               const TwPair<T> *__restrict__ twf, const TwPair<T> *__restrict__ twi,
                                               const ModParams<T> &P, uint32_t nsub, T *lds) {
        constexpr uint32_t CMT = FULL & ~TOP;
        const uint32_t tid = threadIdx.x;
        const TwPair<T> w14 = twf[1];   // forward stage on bit 14 (the inverse one rides in Bfly::inv_norm: P.last_w)
        for (uint32_t tile = blockIdx.x; tile < nsub; tile += gridDim.x) {
            uint32_t tidv = tid;
            asm volatile("" : "+v"(tidv));
            T *base = lhs + ((size_t)tile << (LOGH + 1));
            const T *rb = rhs_ntt + ((size_t)tile << (LOGH + 1));
            T a[E], b[E];
            HB::template gather_tile<TOP>(a, (const T *)base, pdep<CMT>(tidv) * (uint32_t)sizeof(T));
            HB::template gather_tile<TOP>(b, (const T *)(base + ((size_t)1 << LOGH)), pdep<CMT>(tidv) * (uint32_t)sizeof(T));
#pragma unroll
            for (int j = 0; j < E; ++j) {
                a[j] = Bfly<T, CLS>::load_fix(a[j], P);
                b[j] = Bfly<T, CLS>::load_fix(b[j], P);
                Bfly<T, CLS>::template fwd<true, true>(a[j], b[j], w14.w, w14.ws, P);
                if constexpr (Bfly<T, CLS>::IS_FP) {  // the halves' reduction schedules assume inputs no larger than canonical ones
                    a[j] = Bfly<T, CLS>::reduce(a[j], P);
                    b[j] = Bfly<T, CLS>::reduce(b[j], P);
                }
            }
            uint32_t t2 = tidv;
            asm volatile("" : "+v"(t2));
            half(a, lds, tidv, rb, twf, twi, P, 0u);
            half(b, lds, t2, rb + ((size_t)1 << LOGH), twf, twi, P, 1u << LOGH);
            uint32_t t3 = tid;
            asm volatile("" : "+v"(t3));
#pragma unroll
            for (int j = 0; j < E; ++j) {
                if constexpr (Bfly<T, CLS>::IS_FP) {  // whatever the halves' schedules left: back to |v| <= p/2 before the last stage
                    a[j] = Bfly<T, CLS>::reduce(a[j], P);
                    b[j] = Bfly<T, CLS>::reduce(b[j], P);
                }
                Bfly<T, CLS>::inv_norm(a[j], b[j], mul_inv_params<T, CLS>(P));   // (lazy class: constants times 2^B, mul_fused)
                a[j] = Bfly<T, CLS>::finish_inv(a[j], P);
                b[j] = Bfly<T, CLS>::finish_inv(b[j], P);
            }
            HB::template scatter_tile<TOP>(a, base, pdep<CMT>(t3) * (uint32_t)sizeof(T));
            HB::template scatter_tile<TOP>(b, base + ((size_t)1 << LOGH), pdep<CMT>(t3) * (uint32_t)sizeof(T));
        }
    }
};

template <class T, int CLS, int WPW>
__global__ __launch_bounds__((Mul32k<T, CLS>::WPB), WPW) void mul_kernel_32k(T *__restrict__ lhs, const T *__restrict__ rhs_ntt,
                                                                         const TwPair<T> *__restrict__ twf,
                                                                         const TwPair<T> *__restrict__ twi, const ModParams<T> P,
                                                                         uint32_t nsub) {
    using K = Mul32k<T, CLS>;
    __shared__ __attribute__((aligned(16))) T lds[K::HB::LDS_WORDS_1];
    K::run(lhs, rhs_ntt, twf, twi, P, nsub, lds);
}

// -------------------------------------------------------------------------------------------------
// Fused mul_accumulate chain on the wave-block walk (the large-N counterpart of ExtWp, ntt_kernel.hpp):
//     for each o < NOUT:  out[b][o] (+)= inv( sum_{j < J} fwd(terms[b][j]) (.) key_ntt[j][o] )
// i.e. the caller's  for j { plan.fwd(t_j); for o { plan.mul_accumulate(acc_o, t_j, key[j][o]) } }  for o { plan.inv(acc_o) }
// (src/prime64.rs:794, :1085-1128, :872) with every intermediate in registers: (J + NOUT) N words of HBM traffic per batch
// element instead of (2J + 3 J NOUT + 2 NOUT) N.  One batch element per workgroup; the NOUT accumulators live in the NTT
// domain's block layout (each wavefront accumulates its own 1024-word block), `key_ntt` is read in that layout from L2.
// Ordinary loads throughout (no asynchronous prefetch: the accumulators fill the register file).
// -------------------------------------------------------------------------------------------------
template <class T, int LOGN, int CLS, int NOUT, int TWC = 1>
struct ExtBlk {
    using F = NttBlk<T, LOGN, false, CLS, TWC>;
    using I = NttBlk<T, LOGN, true, CLS, TWC>;
    using FB = typename F::B;
    static constexpr int E = F::E, WPB = F::WPB;
    static constexpr uint32_t FULL = F::FULL, RMIO = F::RM0, RMM = F::RM3;
    static_assert(RMM == I::RM0 && RMIO == I::RM3, "forward and inverse schedules must mirror each other");

    static __device__ __forceinline__ void mul_acc_key(T (&acc)[E], const T (&r)[E], const T *__restrict__ key, uint32_t ebase,
                                                       const ModParams<T> &P) {
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

    static __device__ __forceinline__ void run(T *__restrict__ out, const T *__restrict__ terms, const T *__restrict__ key_ntt,
                                               const TwPair<T> *__restrict__ twf, const TwPair<T> *__restrict__ twi,
                                               const ModParams<T> &P, uint32_t nb, uint32_t nterms, bool accumulate, uint32_t ostride,
                                               T *lds) {
        constexpr uint32_t CMIO = FULL & ~RMIO, CMM = FULL & ~RMM;
        const uint32_t tid = threadIdx.x;
        for (uint32_t b = blockIdx.x; b < nb; b += gridDim.x) {
            uint32_t tidv = tid;
            asm volatile("" : "+v"(tidv));
            const T *tb = terms + (((size_t)b * nterms) << LOGN);
            T acc[NOUT][E];
            static_for<0, NOUT>([&](auto o) {
#pragma unroll
                for (int e = 0; e < E; ++e) acc[o.value][e] = 0;
            });
            for (uint32_t j = 0; j < nterms; ++j) {
                // (addresses recomputed per term from an opaque copy of the thread index: hoisted out of this loop they
                // would all be live across it, next to the accumulators, and spill)
                uint32_t tj = tidv;
                asm volatile("" : "+v"(tj));
                T r[E];
                FB::template gather_tile<RMIO>(r, tb + ((size_t)j << LOGN), pdep<CMIO>(tj) * (uint32_t)sizeof(T));
#pragma unroll
                for (int e = 0; e < E; ++e) r[e] = Bfly<T, CLS>::load_fix(r[e], P);
                F::transform(r, lds, tj, twf, P);   // (its first exchange waits for everybody's last read of the buffer)
                if constexpr (mul_fwd_fin<T, CLS>()) {
#pragma unroll
                    for (int e = 0; e < E; ++e) r[e] = Bfly<T, CLS>::finish_fwd(r[e], P);
                }
                if constexpr (Bfly<T, CLS>::IS_FP) {
#pragma unroll
                    for (int e = 0; e < E; ++e) r[e] = Bfly<T, CLS>::reduce(r[e], P);
                }
                static_for<0, NOUT>([&](auto o) {
                    mul_acc_key(acc[o.value], r, key_ntt + (((size_t)j * ostride + o.value) << LOGN), pdep<CMM>(tj), P);
                    if constexpr (Bfly<T, CLS>::IS_FP) {  // every product adds at most 0.875 p to the lazy accumulator
                        if ((j + 1u) % (uint32_t)Bfly<T, CLS>::ACC_REDUCE_EVERY == 0u) {
#pragma unroll
                            for (int e = 0; e < E; ++e) acc[o.value][e] = Bfly<T, CLS>::reduce(acc[o.value][e], P);
                        }
                    }
                });
            }
            static_for<0, NOUT>([&](auto o) {
                T(&a)[E] = acc[o.value];
#pragma unroll
                for (int e = 0; e < E; ++e) a[e] = chain_pre_inverse<T, CLS>(a[e], P);  // lazy accumulator -> inverse input
                // the inverse starts by writing the wavefront's own block: everybody must have finished the previous
                // output's cross-wave gather (or the last forward transform never left the block: harmless extra wait)
                F::template xsync<false>();
                I::template transform<false>(a, lds, tidv, twi, P);
#pragma unroll
                for (int e = 0; e < E; ++e) a[e] = Bfly<T, CLS>::finish_inv(a[e], P);
                T *dst = out + (((size_t)b * ostride + o.value) << LOGN);
                if (accumulate) {
                    T old[E];
                    FB::template gather_tile<RMIO>(old, (const T *)dst, pdep<CMIO>(tidv) * (uint32_t)sizeof(T));
#pragma unroll
                    for (int e = 0; e < E; ++e) a[e] = add_mod<T>(old[e], a[e], P.p);
                }
                FB::template scatter_tile<RMIO>(a, dst, pdep<CMIO>(tidv) * (uint32_t)sizeof(T));
            });
        }
    }
};

template <class T, int LOGN, int CLS, int WPW, int NOUT, int TWC = 1>
__global__ __launch_bounds__((ExtBlk<T, LOGN, CLS, NOUT>::WPB), WPW) void ext_kernel_blk(T *__restrict__ out, const T *__restrict__ terms,
                                                                                   const T *__restrict__ key_ntt,
                                                                                   const TwPair<T> *__restrict__ twf,
                                                                                   const TwPair<T> *__restrict__ twi, const ModParams<T> P,
                                                                                   uint32_t nb, uint32_t nterms, uint32_t accumulate,
                                                                                   uint32_t ostride) {
    using K = ExtBlk<T, LOGN, CLS, NOUT, TWC>;
    __shared__ __attribute__((aligned(16))) T lds[K::FB::LDS_WORDS_1];
    K::run(out, terms, key_ntt, twf, twi, P, nb, nterms, accumulate != 0, ostride, lds);
}

}  // namespace cntt
