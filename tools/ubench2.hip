// Second microbenchmark: SGPR-operand forms, carry chains, selects, and whole-butterfly loops built
// from the product's own Bfly<> code (ntt_arith.hpp), to find what bounds the NTT kernel's VALU time.
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <vector>

#include "../concrete-ntt_amd/csrc/ntt_arith.hpp"

#define CHAINS 8
#define UNROLL 4
using namespace cntt;

enum { T_ADD_S = 0, T_MAD_S, T_CND_VCC, T_CND_SGPR, T_CMP64_CND, T_SUBCO, T_MIN, T_MULLO_S, T_CND_E64VCC, T_ADDCO_ONLY, NT };
static const char *NAMES[NT] = {"v_add_u32 v,s,v", "v_mad_u64_u32 v,s,v", "v_cndmask vcc", "v_cndmask s[..]",
                                "v_cmp_lt_u64+2cnd", "v_sub_co+v_subb_co", "v_min_u32", "v_mul_lo_u32 v,s",
                                "v_cndmask_e64 vcc", "v_add_co_u32 only"};

template <int OP> __global__ void bench(uint64_t *sink, int iters, uint32_t sval, uint64_t smask) {
    uint32_t a = threadIdx.x * 2654435761u + 12345u, b = blockIdx.x * 40503u + 977u;
    uint64_t acc[CHAINS];
    uint32_t acc32[CHAINS];
#pragma unroll
    for (int c = 0; c < CHAINS; ++c) {
        acc[c] = a + c;
        acc32[c] = b + c;
    }
    asm volatile("v_cmp_lt_u32 vcc, %0, %1" ::"v"(a), "v"(b) : "vcc");
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) {
#pragma unroll
            for (int c = 0; c < CHAINS; ++c) {
                if constexpr (OP == T_ADD_S) asm volatile("v_add_u32 %0, %1, %0" : "+v"(acc32[c]) : "s"(sval));
                if constexpr (OP == T_MAD_S) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(acc[c]) : "s"(sval), "v"(b) : "vcc");
                if constexpr (OP == T_CND_VCC) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(acc32[c]) : "v"(a));
                if constexpr (OP == T_CND_SGPR) asm volatile("v_cndmask_b32 %0, %0, %1, %2" : "+v"(acc32[c]) : "v"(a), "s"(smask));
                if constexpr (OP == T_CMP64_CND) {
                    uint32_t lo = (uint32_t)acc[c], hi = (uint32_t)(acc[c] >> 32);
                    asm volatile("v_cmp_lt_u64 vcc, %2, %3\n\tv_cndmask_b32 %0, %0, %4, vcc\n\tv_cndmask_b32 %1, %1, %5, vcc"
                                 : "+v"(lo), "+v"(hi) : "v"(acc[c]), "v"(acc[(c + 1) % CHAINS]), "v"(a), "v"(b) : "vcc");
                    acc[c] = ((uint64_t)hi << 32) | lo;
                }
                if constexpr (OP == T_SUBCO) {
                    uint32_t lo = (uint32_t)acc[c], hi = (uint32_t)(acc[c] >> 32);
                    asm volatile("v_sub_co_u32 %0, vcc, %0, %2\n\tv_subb_co_u32 %1, vcc, %1, %3, vcc" : "+v"(lo), "+v"(hi) : "v"(a), "v"(b) : "vcc");
                    acc[c] = ((uint64_t)hi << 32) | lo;
                }
                if constexpr (OP == T_MIN) asm volatile("v_min_u32 %0, %0, %1" : "+v"(acc32[c]) : "v"(a));
                if constexpr (OP == T_MULLO_S) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(acc32[c]) : "s"(sval));
                if constexpr (OP == T_CND_E64VCC) asm volatile("v_cndmask_b32_e64 %0, %0, %1, vcc" : "+v"(acc32[c]) : "v"(a));
                if constexpr (OP == T_ADDCO_ONLY) asm volatile("v_add_co_u32 %0, vcc, %0, %1" : "+v"(acc32[c]) : "v"(a) : "vcc");
            }
        }
    }
    uint64_t s = 0;
#pragma unroll
    for (int c = 0; c < CHAINS; ++c) s += acc[c] + acc32[c];
    sink[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = s;
}

// whole butterflies: 8 independent (x, y) pairs per lane, twiddles from SGPRs (kernel args) or VGPRs
template <class T, int CLS, bool INV, bool VGPR_TW> __global__ void bfly_bench(T *sink, int iters, ModParams<T> P, T w, T ws) {
    T x[CHAINS], y[CHAINS];
#pragma unroll
    for (int c = 0; c < CHAINS; ++c) {
        x[c] = (T)(threadIdx.x * 977u + c) % P.p;
        y[c] = (T)(blockIdx.x * 131u + 7 * c) % P.p;
    }
    T wv = w, wsv = ws;
    if constexpr (VGPR_TW) {
        wv += threadIdx.x & 1;
        wsv += threadIdx.x & 1;
        asm volatile("" : "+v"(wv), "+v"(wsv));
    }
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int c = 0; c < CHAINS; ++c) {
            if constexpr (INV)
                Bfly<T, CLS>::inv(x[c], y[c], wv, wsv, P);
            else
                Bfly<T, CLS>::fwd(x[c], y[c], wv, wsv, P);
        }
    }
    T s = 0;
#pragma unroll
    for (int c = 0; c < CHAINS; ++c) s += x[c] + y[c];
    sink[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <class F> static float timeit(F launch) {
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    launch(8);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    launch(0);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms = 0;
    (void)hipEventElapsedTime(&ms, e0, e1);
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    return ms;
}

template <int OP> static void run(uint64_t *sink) {
    const int iters = 4096;
    for (int wps : {1, 2, 4}) {
        const int threads = 256 * wps, blocks = 256;
        float ms = timeit([&](int it) {
            hipLaunchKernelGGL(bench<OP>, dim3(blocks), dim3(threads), 0, 0, sink, it ? it : iters, 12345u, 0x5555aaaa5555aaaaull);
        });
        const double n = (double)blocks * threads * iters * UNROLL * CHAINS;
        const int per = (OP == T_CMP64_CND) ? 3 : (OP == T_SUBCO ? 2 : 1);
        printf("%-22s waves/SIMD=%d  wall=%7.3f ms  chip lane-ops= %6.2f T/s (x%d instr)\n", NAMES[OP], wps, ms,
               n * per / (ms * 1e-3) / 1e12, per);
    }
}

template <class T, int CLS, bool INV, bool VG> static void run_bfly(void *sink, const char *name, T p) {
    ModParams<T> P{};
    P.p = p;
    P.neg_p = (T)0 - p;
    P.two_p = (T)(2 * p);
    P.neg_two_p = (T)0 - (T)(2 * p);
    P.pinv_neg = 12345;
    const int iters = 2048;
    for (int wps : {1, 2, 4}) {
        const int threads = 256 * wps, blocks = 256;
        float ms = timeit([&](int it) {
            hipLaunchKernelGGL((bfly_bench<T, CLS, INV, VG>), dim3(blocks), dim3(threads), 0, 0, (T *)sink, it ? it : iters, P, (T)(p / 3), (T)(p / 5));
        });
        const double n = (double)blocks * threads * iters * CHAINS;
        printf("%-34s waves/SIMD=%d  wall=%7.3f ms  %7.1f G butterflies/s\n", name, wps, ms, n / (ms * 1e-3) / 1e9);
    }
}

int main() {
    uint64_t *sink;
    (void)hipMalloc(&sink, 256 * 1024 * sizeof(uint64_t));
    run<T_ADD_S>(sink);
    run<T_MAD_S>(sink);
    run<T_MULLO_S>(sink);
    run<T_MIN>(sink);
    run<T_CND_VCC>(sink);
    run<T_CND_E64VCC>(sink);
    run<T_CND_SGPR>(sink);
    run<T_CMP64_CND>(sink);
    run<T_SUBCO>(sink);
    run<T_ADDCO_ONLY>(sink);
    const uint64_t p62 = 4611686018427322369ull;
    run_bfly<uint64_t, CLS_LAZY, false, false>(sink, "u64 lazy fwd bfly, SGPR twiddle", p62);
    run_bfly<uint64_t, CLS_LAZY, false, true>(sink, "u64 lazy fwd bfly, VGPR twiddle", p62);
    run_bfly<uint64_t, CLS_LAZY, true, true>(sink, "u64 lazy inv bfly, VGPR twiddle", p62);
    run_bfly<uint64_t, CLS_GENERIC, false, true>(sink, "u64 montgomery fwd bfly, VGPR tw", p62);
    run_bfly<uint32_t, CLS_LAZY, false, true>(sink, "u32 lazy fwd bfly, VGPR twiddle", 1062862849u);
    run_bfly<uint32_t, CLS_LAZY, true, true>(sink, "u32 lazy inv bfly, VGPR twiddle", 1062862849u);
    (void)hipFree(sink);
    return 0;
}
