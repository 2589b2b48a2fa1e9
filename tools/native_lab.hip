// Timing lab for the whole-product kernel with the accumulating CRT (native_fused.hpp), shapes of C3 / C5.
// Random tables: timing only.  -DLAB_NOBAR / LAB_UNITW / LAB_NOLDS: timing-only ablations (patched header copies).
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "native_fused.hpp"   // (tools/native_lab.sh puts a patched copy of the kernel headers on the include path)
using namespace cntt;
#ifndef LAB_WPS
#define LAB_WPS 4
#endif
#ifndef LAB_OPT
#define LAB_OPT 0
#endif

template <class F> static float timeit(F launch, int reps) {
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    float wms = 0;
    (void)hipEventRecord(e0);
    while (wms < 1500.f) {
        for (int i = 0; i < 10; ++i) launch();
        (void)hipEventRecord(e1);
        (void)hipEventSynchronize(e1);
        (void)hipEventElapsedTime(&wms, e0, e1);
    }
    (void)hipEventRecord(e0);
    for (int i = 0; i < reps; ++i) launch();
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms = 0;
    (void)hipEventElapsedTime(&ms, e0, e1);
    return ms / reps;
}
static const uint32_t PR[10] = {1062862849u, 1063059457u, 1064697857u, 1065484289u, 1068236801u,
                                1068433409u, 1068564481u, 1069219841u, 1071513601u, 1073479681u};

template <int KIND, int LOGN> static void run(const char *tag, uint32_t batch) {
    constexpr int KP = NativeShape<KIND>::KP;
    const size_t n = (size_t)1 << LOGN, words = n * batch;
    uint64_t *lhs, *rhs, *prod;
    (void)hipMalloc(&lhs, words * 8);
    (void)hipMalloc(&rhs, words * 8);
    (void)hipMalloc(&prod, words * 8);
    std::vector<uint64_t> h(words);
    uint64_t s = 88172645463325252ull;
    for (auto &v : h) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; v = s; }
    (void)hipMemcpy(lhs, h.data(), words * 8, hipMemcpyHostToDevice);
    if (NativeShape<KIND>::BINARY) for (auto &v : h) v &= 1;
    (void)hipMemcpy(rhs, h.data(), words * 8, hipMemcpyHostToDevice);
    FusedTables<KP> F{};
    SplitArgs S{};
    AccArgs C{};
    for (int i = 0; i < KP; ++i) {
        const uint32_t p = PR[i];
        std::vector<TwPair<uint32_t>> t(n);
        for (auto &e : t) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; e.w = (uint32_t)(s % p); e.ws = (uint32_t)(((uint64_t)e.w << 32) / p); }
        TwPair<uint32_t> *df, *di;
        (void)hipMalloc(&df, n * 8);
        (void)hipMalloc(&di, n * 8);
        (void)hipMemcpy(df, t.data(), n * 8, hipMemcpyHostToDevice);
        (void)hipMemcpy(di, t.data(), n * 8, hipMemcpyHostToDevice);
        F.twf[i] = df; F.twi[i] = di;
        ModParams<uint32_t> &P = F.P[i];
        P.p = p; P.neg_p = 0u - p; P.two_p = 2 * p; P.neg_two_p = 0u - 2 * p; P.big_q = 30;
        P.p_barrett = (uint32_t)((((uint64_t)1) << 61) / p);
        P.n_inv = 12345; P.n_inv_shoup = (uint32_t)(((uint64_t)12345 << 32) / p);
        P.last_w = 54321; P.last_w_shoup = (uint32_t)(((uint64_t)54321 << 32) / p);
        S.prime[i] = p; S.c[i] = (uint32_t)((((uint64_t)1) << 32) % p); S.one_shoup[i] = (uint32_t)((((uint64_t)1) << 32) / p);
        C.c_lo[i] = s; C.f[i] = (uint32_t)((((uint64_t)1) << 59) / p); C.m60[i] = (uint32_t)((((uint64_t)1) << 60) / p);
    }
    using K0 = NttKernel<uint32_t, LOGN, false, CLS_LAZY, false, 1>;
    constexpr int BLK = 256, PPB = BLK / K0::TPP;
    const uint32_t grid = (batch + PPB - 1) / PPB;
    const float ms = timeit([&] {
        hipLaunchKernelGGL((native_polymul_kernel_acc<KIND, LOGN, BLK, LAB_WPS, LAB_OPT>), dim3(grid), dim3(BLK), 0, 0, prod, lhs, rhs, F, S, C, batch);
    }, 10);
    printf("%s wps=%d opt=%d  %.4f ms  %s\n", tag, LAB_WPS, LAB_OPT, ms, hipGetErrorString(hipGetLastError()));
    fflush(stdout);
}
int main() {
    run<1, 12>("C3 native64 N=4096 x16384", 16384);
    run<4, 11>("C5 native_binary64 N=2048 x65536", 65536);
    return 0;
}
