// Ablation lab for the headline kernel (prime64 N=1024, lazy class): where does the time go?
// Variants share the product's kernel source (ntt_kernel.hpp):
//   baseline kernel with LAB flags: 1 no per-thread twiddle loads, 2 no LDS exchange, 4 no global
//   load/store; persistent software-pipelined kernel (ntt_kernel_wp).
// Every row reports HIP-event time, the in-kernel shader clock (s_memtime / s_memrealtime) and the
// algorithmic-bytes rate; persistent variants are checked bit for bit against the baseline.
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <vector>

#include "../concrete-ntt_amd/csrc/ntt_kernel.hpp"
using namespace cntt;

static uint64_t *g_data;
static TwPair<uint64_t> *g_tw;
static ModParams<uint64_t> g_P;
static const uint32_t BATCH = 65536;
static const int REPS = 200;
#ifndef LAB_WPW
#define LAB_WPW 3  // waves per SIMD the persistent kernel is compiled for (= workgroups of 256 threads per CU)
#endif

static unsigned long long *g_stamp;  // device buffer, 2 slots per wave
static const size_t STAMP_WAVES = 65536 + 1024;
static void clk_reset() { (void)hipMemset(g_stamp, 0, STAMP_WAVES * 16); }
static double clk_read_mhz() {
    std::vector<unsigned long long> hst(STAMP_WAVES * 2);
    (void)hipMemcpy(hst.data(), g_stamp, STAMP_WAVES * 16, hipMemcpyDeviceToHost);
    double t = 0, r = 0;
    for (size_t w = 0; w < STAMP_WAVES; ++w) {
        t += (double)hst[2 * w];
        r += (double)hst[2 * w + 1];
    }
    return r > 0 ? 100.0 * t / r : 0;
}

template <class F> static void timeit(const char *name, F launch) {
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    {   // steady state: ~1.5 s of back-to-back launches before the timed region (DVFS ramp)
        hipEvent_t w0, w1;
        (void)hipEventCreate(&w0);
        (void)hipEventCreate(&w1);
        float wms = 0;
        (void)hipEventRecord(w0);
        while (wms < 1500.f) {
            for (int i = 0; i < 200; ++i) launch();
            (void)hipEventRecord(w1);
            (void)hipEventSynchronize(w1);
            (void)hipEventElapsedTime(&wms, w0, w1);
        }
        (void)hipEventDestroy(w0);
        (void)hipEventDestroy(w1);
    }
    clk_reset();
    (void)hipEventRecord(e0);
    for (int i = 0; i < REPS; ++i) launch();
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms = 0;
    (void)hipEventElapsedTime(&ms, e0, e1);
    ms /= REPS;
    const hipError_t err = hipGetLastError();
    printf("%-30s %8.1f us  %7.1f M NTT/s  %5.1f %% of 8 TB/s  clock %6.0f MHz %s\n", name, ms * 1e3,
           BATCH / (ms * 1e-3) / 1e6, 100.0 * BATCH * 16384.0 / (ms * 1e-3) / 8e12, clk_read_mhz(),
           err == hipSuccess ? "" : hipGetErrorString(err));
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
}

template <int LAB0, bool INV> static void base(const char *name) {
    constexpr int LAB = LAB0 | 8;
    using K = NttKernel<uint64_t, 10, INV, CLS_LAZY, false, LAB>;
    const uint32_t grid = (BATCH + K::PPB - 1) / K::PPB;
    timeit(name, [&] {
        hipLaunchKernelGGL((ntt_kernel<uint64_t, 10, INV, CLS_LAZY, false, LAB>), dim3(grid), dim3(K::BLOCK), 0, 0, g_data, g_tw, g_P, BATCH, 0u);
    });
}
// STAMP (in-kernel clock stamps) only where the kernel still compiles without spills: the persistent kernel's
// inline-asm prefetch must never meet a spilled register (see ntt_kernel.hpp gather_async).
template <int WPB, int WPW, bool INV, bool STAMP = !INV> static void wp(const char *name, int blocks_per_cu) {
    using K = NttWp<uint64_t, 10, INV, CLS_LAZY, WPB, STAMP>;
    const uint32_t ntiles = (BATCH + K::PPB - 1) / K::PPB;
    uint32_t grid = 256u * (uint32_t)blocks_per_cu;
    if (grid > ntiles) grid = ntiles;
    timeit(name, [&] {
        hipLaunchKernelGGL((ntt_kernel_wp<uint64_t, 10, INV, CLS_LAZY, WPB, WPW, STAMP>), dim3(grid), dim3(WPB), 0, 0, g_data, g_tw, g_P, BATCH);
    });
}

template <int WPB, int WPW, bool INV> static bool check_wp(const std::vector<uint64_t> &src, uint32_t batch, uint32_t grid) {
    uint64_t *a, *b;
    const size_t bytes = (size_t)batch * 1024 * 8;
    (void)hipMalloc(&a, bytes);
    (void)hipMalloc(&b, bytes);
    (void)hipMemcpy(a, src.data(), bytes, hipMemcpyHostToDevice);
    (void)hipMemcpy(b, src.data(), bytes, hipMemcpyHostToDevice);
    using K0 = NttKernel<uint64_t, 10, INV, CLS_LAZY, false, 0>;
    hipLaunchKernelGGL((ntt_kernel<uint64_t, 10, INV, CLS_LAZY, false, 0>), dim3((batch + K0::PPB - 1) / K0::PPB), dim3(K0::BLOCK), 0, 0, a, g_tw, g_P, batch, 0u);
    hipLaunchKernelGGL((ntt_kernel_wp<uint64_t, 10, INV, CLS_LAZY, WPB, WPW, false>), dim3(grid), dim3(WPB), 0, 0, b, g_tw, g_P, batch);
    std::vector<uint64_t> ha((size_t)batch * 1024), hb((size_t)batch * 1024);
    (void)hipMemcpy(ha.data(), a, bytes, hipMemcpyDeviceToHost);
    (void)hipMemcpy(hb.data(), b, bytes, hipMemcpyDeviceToHost);
    (void)hipFree(a);
    (void)hipFree(b);
    size_t bad = 0;
    for (size_t i = 0; i < ha.size(); ++i) bad += ha[i] != hb[i];
    return bad == 0;
}

int main() {
    const uint32_t n = 1024;
    const uint64_t p = 4611686018427322369ull;
    (void)hipMalloc(&g_data, (size_t)BATCH * n * 8);
    (void)hipMalloc(&g_tw, n * sizeof(TwPair<uint64_t>));
    (void)hipMalloc(&g_stamp, STAMP_WAVES * 16);
    (void)hipMemcpyToSymbol(HIP_SYMBOL(cntt_lab_buf), &g_stamp, sizeof g_stamp);
    std::vector<uint64_t> h((size_t)BATCH * n);
    uint64_t s = 88172645463325252ull;
    for (auto &v : h) {
        s ^= s << 13; s ^= s >> 7; s ^= s << 17;
        v = s % p;
    }
    (void)hipMemcpy(g_data, h.data(), h.size() * 8, hipMemcpyHostToDevice);
    std::vector<TwPair<uint64_t>> ht(n);
    for (uint32_t i = 0; i < n; ++i) {
        s ^= s << 13; s ^= s >> 7; s ^= s << 17;
        ht[i].w = s % p;
        ht[i].ws = (uint64_t)((((unsigned __int128)ht[i].w) << 64) / p);
    }
    (void)hipMemcpy(g_tw, ht.data(), n * sizeof(TwPair<uint64_t>), hipMemcpyHostToDevice);
    g_P = ModParams<uint64_t>{};
    g_P.p = p; g_P.neg_p = 0 - p; g_P.two_p = 2 * p; g_P.neg_two_p = 0 - 2 * p;

    base<0, false>("fwd baseline");
    base<1, false>("fwd no-twiddle-loads");
    base<2, false>("fwd no-LDS-exchange");
    base<4, false>("fwd no-global");
    base<7, false>("fwd ALU only (1+2+4)");
    base<0, true>("inv baseline");
    base<7, true>("inv ALU only (1+2+4)");
    {
        std::vector<uint64_t> src(h.begin(), h.begin() + (size_t)1333 * 1024);
        printf("check wp<256,LAB_WPW> fwd %d inv %d\n", (int)check_wp<256, LAB_WPW, false>(src, 1333, 7),
               (int)check_wp<256, LAB_WPW, true>(src, 1333, 64));
    }
    wp<256, LAB_WPW, false, false>("wp 256thr xLAB_WPW/CU fwd", LAB_WPW);
    wp<256, LAB_WPW, false, false>("wp 256thr x2/CU fwd", 2);
    wp<256, LAB_WPW, true, false>("wp 256thr xLAB_WPW/CU inv", LAB_WPW);
    return 0;
}
