// Ablation lab for the headline kernel (prime64 N=1024, lazy class): where does the time go?
// Everything lab-specific lives HERE; the product headers carry no instrumentation.  The lab composes the product's
// own building blocks (NttKernel::stages / gather / scatter, NttWp::run):
//   * "ALU only": the ten butterfly stages of the product schedule on register-resident data, no global traffic, no
//     LDS exchange, first-stage (uniform) twiddles only -- the VALU ceiling of the transform;
//   * the product's persistent kernel wrapped with shader-clock / real-time stamps (s_memtime / s_memrealtime);
//   * the persistent kernel checked bit for bit against the one-polynomial-per-workgroup kernel.
// Build: tools/lab_build.sh ntt [flags]   |   tools/lab_build.sh xlane (-DCNTT_LAB_XLANE on a patched header copy)
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <vector>

#include "ntt_kernel.hpp"   // the product header, or (ntt_lab_xlane) a scratch copy with tools/xlane_lab.patch applied
using namespace cntt;

static uint64_t *g_data;
static TwPair<uint64_t> *g_tw;
static ModParams<uint64_t> g_P;
static const uint32_t BATCH = 65536;
static const int REPS = 200;
#ifndef LAB_CLS
#define LAB_CLS CLS_LAZY   // -DLAB_CLS=CLS_STRICT -DLAB_P=9223372036853661697ull: the 63-bit class
#endif
#ifndef LAB_P
#define LAB_P 4611686018427322369ull
#endif
#ifndef LAB_WPW
#define LAB_WPW 3  // waves per SIMD the persistent kernel is compiled for (= workgroups of 256 threads per CU)
#endif

// ---- clock stamps: 2 slots per wave -----------------------------------------------------------------------
__device__ unsigned long long *lab_buf;
struct LabStamp {
    unsigned long long t0, r0;
    __device__ __forceinline__ void begin() {
        t0 = __builtin_amdgcn_s_memtime();
        r0 = __builtin_amdgcn_s_memrealtime();
    }
    __device__ __forceinline__ void end() {
        const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
        if ((threadIdx.x & 63) == 0) {
            const size_t w = (size_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
            lab_buf[2 * w] = t1 - t0;
            lab_buf[2 * w + 1] = r1 - r0;
        }
    }
};
static unsigned long long *g_stamp;
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

// ---- lab kernels ---------------------------------------------------------------------------------------------
// every pass's stages on registers (addresses as thread 0 of the polynomial: uniform twiddles, scalar loads)
template <class K, int PASS = 0> __device__ __forceinline__ void all_stages(uint64_t (&r)[K::E], const TwPair<uint64_t> *tw,
                                                                            const ModParams<uint64_t> &P) {
    if constexpr (PASS < K::NPASS) {
        K::template stages<PASS>(r, 0u, 0u, 0u, tw, P);
        all_stages<K, PASS + 1>(r, tw, P);
    }
}
// the same with THREAD-DEPENDENT twiddles (vector-register operands, read from the table in L2 by ordinary loads): what the butterflies cost when
// their multiplicands are not scalar registers
template <class K, int PASS = 0> __device__ __forceinline__ void all_stages_vtw(uint64_t (&r)[K::E], const TwPair<uint64_t> *tw,
                                                                                const ModParams<uint64_t> &P, uint32_t tid) {
    if constexpr (PASS < K::NPASS) {
        constexpr uint32_t CM = K::FULL & ~K::S::RMASK[PASS];
        K::template stages<PASS>(r, pdep<CM>(tid), 0u, 0u, tw, P);
        all_stages_vtw<K, PASS + 1>(r, tw, P, tid);
    }
}
template <bool INV> __global__ __launch_bounds__(256) void lab_alu_vtw(uint64_t *sink, const TwPair<uint64_t> *tw, const ModParams<uint64_t> P) {
    using K = NttKernel<uint64_t, 10, INV, LAB_CLS, false>;
    LabStamp st;
    st.begin();
    uint64_t r[K::E];
#pragma unroll
    for (int j = 0; j < K::E; ++j) r[j] = (uint64_t)(threadIdx.x * 77u + j);
    all_stages_vtw<K>(r, tw, P, threadIdx.x & (K::TPP - 1));
    uint64_t acc = 0;
#pragma unroll
    for (int j = 0; j < K::E; ++j) acc ^= r[j];
    if (acc == 0x1234567ull) sink[0] = acc;
    st.end();
}
template <bool INV> __global__ __launch_bounds__(256) void lab_alu_only(uint64_t *sink, const TwPair<uint64_t> *tw,
                                                                       const ModParams<uint64_t> P) {
    using K = NttKernel<uint64_t, 10, INV, LAB_CLS, false>;
    LabStamp st;
    st.begin();
    uint64_t r[K::E];
#pragma unroll
    for (int j = 0; j < K::E; ++j) r[j] = (uint64_t)(threadIdx.x * 77u + j);
    all_stages<K>(r, tw, P);
    uint64_t acc = 0;  // keep the work alive without stores
#pragma unroll
    for (int j = 0; j < K::E; ++j) acc ^= r[j];
    if (acc == 0x1234567ull) sink[0] = acc;
    st.end();
}
template <bool INV, int WPB, int WPW, int TWC = 0> __global__ __launch_bounds__(WPB, WPW) void lab_wp_stamped(
    uint64_t *data, const TwPair<uint64_t> *tw, const ModParams<uint64_t> P, uint32_t nsub) {
    using K = NttWp<uint64_t, 10, INV, LAB_CLS, WPB>;
    __shared__ __attribute__((aligned(16))) uint64_t lds[(size_t)K::PPB * K::B::LDS_WORDS_1];
    __shared__ __attribute__((aligned(16))) TwPair<uint64_t> img[K::B::IMG_ENTRIES];
    LabStamp st;
    st.begin();
    K::template run<true, TWC>(data, tw, P, nsub, lds, img);
    st.end();
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
    printf("%-34s %8.1f us  %7.1f M NTT/s  %5.1f %% of 8 TB/s  clock %6.0f MHz %s\n", name, ms * 1e3,
           BATCH / (ms * 1e-3) / 1e6, 100.0 * BATCH * 16384.0 / (ms * 1e-3) / 8e12, clk_read_mhz(),
           err == hipSuccess ? "" : hipGetErrorString(err));
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
}

template <bool INV> static void product_kernel(const char *name) {  // what cntt_prime64_{fwd,inv}_batch launches at N=1024
    using W = NttWp<uint64_t, 10, INV, LAB_CLS, 256>;
    const uint32_t ntiles = (BATCH + W::PPB - 1) / W::PPB;
    uint32_t grid = 256u * 3u;
    if (grid > ntiles) grid = ntiles;
    timeit(name, [&] {
        hipLaunchKernelGGL((ntt_kernel_wp<uint64_t, 10, INV, LAB_CLS, 256, 3>), dim3(grid), dim3(256), 0, 0, g_data, g_tw, g_P, BATCH);
    });
}
template <bool INV> static void alu_only(const char *name) {
    using K = NttKernel<uint64_t, 10, INV, LAB_CLS, false>;
    const uint32_t grid = BATCH / (256 / K::TPP);  // as many threads as the real transform of the batch
    timeit(name, [&] { hipLaunchKernelGGL((lab_alu_only<INV>), dim3(grid), dim3(256), 0, 0, g_data, g_tw, g_P); });
}
template <bool INV> static void alu_vtw(const char *name) {
    using K = NttKernel<uint64_t, 10, INV, LAB_CLS, false>;
    const uint32_t grid = BATCH / (256 / K::TPP);
    timeit(name, [&] { hipLaunchKernelGGL((lab_alu_vtw<INV>), dim3(grid), dim3(256), 0, 0, g_data, g_tw, g_P); });
}
template <int WPB, int WPW, bool INV, int TWC = 0> static void wp_stamped(const char *name, int blocks_per_cu) {
    using K = NttWp<uint64_t, 10, INV, LAB_CLS, WPB>;
    const uint32_t ntiles = (BATCH + K::PPB - 1) / K::PPB;
    uint32_t grid = 256u * (uint32_t)blocks_per_cu;
    if (grid > ntiles) grid = ntiles;
    timeit(name, [&] {
        hipLaunchKernelGGL((lab_wp_stamped<INV, WPB, WPW, TWC>), dim3(grid), dim3(WPB), 0, 0, g_data, g_tw, g_P, BATCH);
    });
}

template <bool INV> static bool check_wp(const std::vector<uint64_t> &src, uint32_t batch, uint32_t grid) {
    uint64_t *a, *b;
    const size_t bytes = (size_t)batch * 1024 * 8;
    (void)hipMalloc(&a, bytes);
    (void)hipMalloc(&b, bytes);
    (void)hipMemcpy(a, src.data(), bytes, hipMemcpyHostToDevice);
    (void)hipMemcpy(b, src.data(), bytes, hipMemcpyHostToDevice);
    using K0 = NttKernel<uint64_t, 10, INV, LAB_CLS, false>;
    hipLaunchKernelGGL((ntt_kernel<uint64_t, 10, INV, LAB_CLS, false>), dim3((batch + K0::PPB - 1) / K0::PPB), dim3(K0::BLOCK), 0, 0, a, g_tw, g_P, batch, 0u);
    hipLaunchKernelGGL((ntt_kernel_wp<uint64_t, 10, INV, LAB_CLS, 256, 3>), dim3(grid), dim3(256), 0, 0, b, g_tw, g_P, batch);
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
    const uint64_t p = LAB_P;
    (void)hipMalloc(&g_data, (size_t)BATCH * n * 8);
    (void)hipMalloc(&g_tw, n * sizeof(TwPair<uint64_t>));
    (void)hipMalloc(&g_stamp, STAMP_WAVES * 16);
    (void)hipMemcpyToSymbol(HIP_SYMBOL(lab_buf), &g_stamp, sizeof g_stamp);
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

    product_kernel<false>("fwd product kernel (wp 256x3)");
    product_kernel<true>("inv product kernel (wp 256x3)");
    alu_only<false>("fwd ALU only (stages on registers)");
    alu_only<true>("inv ALU only (stages on registers)");
    alu_vtw<false>("fwd ALU, per-thread twiddles (L2)");
    alu_vtw<true>("inv ALU, per-thread twiddles (L2)");
    {
        std::vector<uint64_t> src(h.begin(), h.begin() + (size_t)1333 * 1024);
        printf("check wp vs one-polynomial-per-workgroup kernel: fwd %d inv %d\n", (int)check_wp<false>(src, 1333, 7),
               (int)check_wp<true>(src, 1333, 64));
    }
    wp_stamped<256, LAB_WPW, false>("fwd wp stamped, 3 WG/CU", 3);
    wp_stamped<256, LAB_WPW, false>("fwd wp stamped, 2 WG/CU", 2);
    wp_stamped<256, LAB_WPW, true>("inv wp stamped, 3 WG/CU", 3);
    // four waves per SIMD (128 VGPRs): twiddle image reads in chunks of 2 / 4 pairs
    wp_stamped<1024, 4, false, 2>("fwd wp 1024thr x1/CU, TWC=2", 1);
    wp_stamped<1024, 4, false, 4>("fwd wp 1024thr x1/CU, TWC=4", 1);
    wp_stamped<512, 4, false, 2>("fwd wp 512thr x2/CU, TWC=2", 2);
    wp_stamped<1024, 4, true, 2>("inv wp 1024thr x1/CU, TWC=2", 1);
    wp_stamped<256, 3, false, 4>("fwd wp 256thr x3/CU, TWC=4", 3);
    return 0;
}
