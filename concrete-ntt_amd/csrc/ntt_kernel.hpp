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
#include "ntt_arith.hpp"

namespace cntt {

template <int BITS, int LOGN, bool INV> struct Sched;  // specialisations: sched_gen.inc
#include "sched_gen.inc"

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
template <class T, int LOGN, bool INV, int CLS, bool SUB>
struct NttKernel {
    static constexpr int BITS = sizeof(T) * 8;
    using S = Sched<BITS, LOGN, INV>;
    static constexpr int LOGE = S::LOGE, E = 1 << LOGE, NPASS = S::NPASS, BLOCK = S::BLOCK;
    static constexpr int TPP = 1 << (LOGN - LOGE);
    static constexpr int PPB = (BLOCK / TPP) > 0 ? (BLOCK / TPP) : 1;
    static constexpr uint32_t FULL = (LOGN >= 32) ? 0xffffffffu : ((1u << LOGN) - 1u);
    static constexpr bool WAVE_PRIVATE = TPP <= 64;  // a polynomial never spans two wavefronts
    static constexpr int MAXV = 16 / (int)sizeof(T);  // elements per 16-byte access
    static constexpr size_t LDS_ELEMS = (NPASS > 1) ? ((size_t)PPB << LOGN) : 1;

    static __device__ __forceinline__ uint32_t phys(uint32_t e) {
        uint32_t o = e;
        if constexpr (S::SWZ_M0 != 0) o ^= ((e >> S::SWZ_SH0) & S::SWZ_M0) << S::SWZ_L0;
        if constexpr (S::SWZ_M1 != 0) o ^= ((e >> S::SWZ_SH1) & S::SWZ_M1) << S::SWZ_L1;
        return o;
    }

    static __device__ __forceinline__ void sync() {
        if constexpr (WAVE_PRIVATE) {
            // same-wave LDS operations execute in issue order; only the compiler must not reorder
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
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
#pragma unroll
        for (int j = 0; j < E; j += NV) {
            const uint32_t e = ebase | cdep((uint32_t)j, RM);
            const uint32_t a = lds ? phys(e) : e;
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
#pragma unroll
        for (int j = 0; j < E; j += NV) {
            const uint32_t e = ebase | cdep((uint32_t)j, RM);
            const uint32_t a = lds ? phys(e) : e;
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
    template <int K, int GI>
    static __device__ __forceinline__ void stage(T (&r)[E], uint32_t ebase, uint32_t qpre, uint32_t depth,
                                                 const TwPair<T> *__restrict__ tw, const ModParams<T> &P) {
        constexpr uint32_t RM = S::RMASK[K], GM = S::GMASK[K], CM = FULL & ~RM;
        constexpr int b = nth_stage_bit(GM, GI);  // element-index bit of this stage
        constexpr int k = crank(RM, b);           // register-index bit that carries it
        constexpr int NHI = 1 << (LOGE - 1 - k);  // register bits above it -> distinct twiddles per thread
        // table entry = 2^(LOGN-1-b) + (e >> (b+1))  [<< depth / + sub-block prefix for SUB kernels]
        uint32_t toff = (1u << (LOGN - 1 - b)) << (SUB ? depth : 0u);
        if constexpr ((CM >> (b + 1)) != 0u) toff += ebase >> (b + 1);
        if constexpr (SUB) toff += qpre >> (b + 1);
        TwPair<T> w[NHI];
#pragma unroll
        for (int h = 0; h < NHI; ++h) {
            w[h] = tw[toff + (cdep((uint32_t)h << (k + 1), RM) >> (b + 1))];
        }
#pragma unroll
        for (int j = 0; j < E; ++j) {
            if ((j >> k) & 1) continue;
            const int h = j >> (k + 1);
            if constexpr (INV)
                Bfly<T, CLS>::inv(r[j], r[j | (1 << k)], w[h].w, w[h].ws, P);
            else
                Bfly<T, CLS>::fwd(r[j], r[j | (1 << k)], w[h].w, w[h].ws, P);
        }
    }

    template <int K, int GI = 0>
    static __device__ __forceinline__ void stages(T (&r)[E], uint32_t ebase, uint32_t qpre, uint32_t depth,
                                                  const TwPair<T> *__restrict__ tw, const ModParams<T> &P) {
        if constexpr (GI < cpop(S::GMASK[K])) {
            stage<K, GI>(r, ebase, qpre, depth, tw, P);
            stages<K, GI + 1>(r, ebase, qpre, depth, tw, P);
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
        T *lds = lds_all + ((size_t)pl << LOGN);
        uint32_t qpre = 0;
        if constexpr (SUB) qpre = (sub & ((1u << depth) - 1u)) << LOGN;
        T r[E];
#pragma unroll
        for (int j = 0; j < E; ++j) r[j] = 0;
        run_pass<0>(r, g, lds, tid, active, qpre, depth, tw, P);
    }
};

template <class T, int LOGN, bool INV, int CLS, bool SUB>
__global__ __launch_bounds__((NttKernel<T, LOGN, INV, CLS, SUB>::BLOCK)) void ntt_kernel(
    T *__restrict__ data, const TwPair<T> *__restrict__ tw, const ModParams<T> P, uint32_t nsub, uint32_t depth) {
    using K = NttKernel<T, LOGN, INV, CLS, SUB>;
    __shared__ __attribute__((aligned(16))) T lds[K::LDS_ELEMS];
    K::run(data, tw, P, nsub, depth, lds);
}

}  // namespace cntt
