// Timing lab for the wave-block kernels (ntt_blk.hpp), prime64 lazy class, N = 4096 / 8192 / 16384.
// Composes the product's own NttBlk::run under shader-clock stamps; -DCNTT_BLK_LAB=<bits> selects a timing-only
// ablation (see ntt_blk.hpp), -DLAB_TWC=<n> the twiddle chunk.  Random twiddles: timing only, no result check here
// (parity is tests/test_gpu_parity.py through the C ABI).
// Build: tools/lab_build.sh blk [-DCNTT_BLK_LAB=1 ...]   (the switches live in tools/blk_lab.patch, not in the product header)
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

// per-wave phase accumulators (shader-clock cycles between consecutive stamps of ntt_blk.hpp), kept in LDS next to the
// 128 KiB exchange buffer and dumped by the wrapper kernel.  -DLAB_STAMPS enables them.
#ifdef LAB_STAMPS
__shared__ unsigned long long lab_acc[16][10];
__shared__ unsigned long long lab_prev[16];
__device__ __forceinline__ void lab_stamp(int k) {
    const unsigned long long t = __builtin_amdgcn_s_memtime();
    if ((threadIdx.x & 63) == 0) {
        const int w = threadIdx.x >> 6;
        if (k != 0 || lab_prev[w] != 0) lab_acc[w][k] += t - lab_prev[w];   // stamp 0 closes the previous iteration's tail
        lab_prev[w] = t;
    }
}
#define CNTT_BLK_STAMP(k) lab_stamp(k)
#endif
#include "ntt_blk.hpp"   // a scratch copy with tools/blk_lab.patch applied (tools/lab_build.sh)
using namespace cntt;
__device__ unsigned long long lab_phase[10];

#ifndef LAB_TWC
#define LAB_TWC 2
#endif
#ifndef LAB_TWC_F
#define LAB_TWC_F LAB_TWC
#endif
#ifndef LAB_TWC_I
#define LAB_TWC_I LAB_TWC
#endif
#ifndef LAB_CLS
#define LAB_CLS CLS_LAZY
#endif

__device__ unsigned long long *lab_buf;
static unsigned long long *g_stamp;
static const size_t STAMP_WAVES = 1 << 16;

template <int LOGN, bool INV, int WPW, int TWC>
__global__ __launch_bounds__((NttBlk<uint64_t, LOGN, INV, LAB_CLS>::WPB), WPW) void lab_blk(uint64_t *data, const TwPair<uint64_t> *tw,
                                                                                          const ModParams<uint64_t> P, uint32_t nsub) {
    using K = NttBlk<uint64_t, LOGN, INV, LAB_CLS, TWC>;
    __shared__ __attribute__((aligned(16))) uint64_t lds[K::B::LDS_WORDS_1];
#ifdef LAB_STAMPS
    if (threadIdx.x < 16) {
        lab_prev[threadIdx.x] = 0;
        for (int k = 0; k < 10; ++k) lab_acc[threadIdx.x][k] = 0;
    }
    __syncthreads();
#endif
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    K::run(data, tw, P, nsub, lds);
#ifdef LAB_STAMPS
    __syncthreads();
    if (threadIdx.x < 10) {
        unsigned long long sum = 0;
        for (int w = 0; w < (int)(blockDim.x >> 6); ++w) sum += lab_acc[w][threadIdx.x];
        atomicAdd(&lab_phase[threadIdx.x], sum);
    }
#endif
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    if ((threadIdx.x & 63) == 0) {
        const size_t w = ((size_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)) & (STAMP_WAVES - 1);
        lab_buf[2 * w] = t1 - t0;
        lab_buf[2 * w + 1] = r1 - r0;
    }
}

static double clk_mhz() {
    std::vector<unsigned long long> h(STAMP_WAVES * 2);
    (void)hipMemcpy(h.data(), g_stamp, STAMP_WAVES * 16, hipMemcpyDeviceToHost);
    double t = 0, r = 0;
    for (size_t w = 0; w < STAMP_WAVES; ++w) {
        t += (double)h[2 * w];
        r += (double)h[2 * w + 1];
    }
    return r > 0 ? 100.0 * t / r : 0;
}

template <class F> static float timeit(F launch, int reps) {
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    float wms = 0;
    (void)hipEventRecord(e0);
    while (wms < 1200.f) {  // DVFS ramp: steady state before the timed region
        for (int i = 0; i < 20; ++i) launch();
        (void)hipEventRecord(e1);
        (void)hipEventSynchronize(e1);
        (void)hipEventElapsedTime(&wms, e0, e1);
    }
    (void)hipMemset(g_stamp, 0, STAMP_WAVES * 16);
    (void)hipEventRecord(e0);
    for (int i = 0; i < reps; ++i) launch();
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms = 0;
    (void)hipEventElapsedTime(&ms, e0, e1);
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    return ms / reps;
}

template <int LOGN, bool INV, int WPW, int TWC> static void run_one(uint64_t *data, const TwPair<uint64_t> *tw, const ModParams<uint64_t> &P,
                                                                    size_t bytes, int bpc) {
    using K = NttBlk<uint64_t, LOGN, INV, LAB_CLS, TWC>;
    const uint32_t nsub = (uint32_t)(bytes >> (LOGN + 3));
    uint32_t grid = 256u * (uint32_t)bpc;
    if (grid > nsub) grid = nsub;
    const float ms = timeit([&] { hipLaunchKernelGGL((lab_blk<LOGN, INV, WPW, TWC>), dim3(grid), dim3(K::WPB), 0, 0, data, tw, P, nsub); }, 20);
    const hipError_t err = hipGetLastError();
#ifdef LAB_STAMPS
    {   // one more launch with clean accumulators: average cycles per wave per polynomial in each phase
        unsigned long long z[10] = {0}, h[10];
        (void)hipMemcpyToSymbol(HIP_SYMBOL(lab_phase), z, sizeof z);
        hipLaunchKernelGGL((lab_blk<LOGN, INV, WPW, TWC>), dim3(grid), dim3(K::WPB), 0, 0, data, tw, P, nsub);
        (void)hipDeviceSynchronize();
        (void)hipMemcpyFromSymbol(h, HIP_SYMBOL(lab_phase), sizeof h);
        const double den = (double)nsub * (K::WPB / 64);
        static const char *nm[10] = {"loop tail -> top", "pass 0 (after the prefetch issue)", "exchange 0 (2 barriers)", "pass 1", "exch1 + pass 2", "exch2 + pass 3",
                                     "store issue", "prefetch wait + unpack", "top -> prefetch issued", "finish + transpose"};
        double tot = 0;
        for (int k = 0; k < 10; ++k) tot += (double)h[k];
        for (int k = 0; k < 10; ++k) printf("    phase %d %-34s %9.0f cycles/wave/poly  %5.1f %%\n", k, nm[k], h[k] / den, 100.0 * h[k] / tot);
    }
#endif
    printf("lab=%d cls=%d n=%5d %s twc=%d wg/cu=%d  %8.2f ns/poly  %5.1f %% of 8 TB/s  clock %5.0f MHz %s\n", CNTT_BLK_LAB, LAB_CLS, 1 << LOGN,
           INV ? "inv" : "fwd", TWC, bpc, ms * 1e6 / nsub, 100.0 * 2.0 * bytes / (ms * 1e-3) / 8e12, clk_mhz(),
           err == hipSuccess ? "" : hipGetErrorString(err));
    fflush(stdout);
}

int main() {
    const uint64_t p = 4611686018427322369ull;
    const size_t bytes = (size_t)1 << 30;
    uint64_t *data;
    TwPair<uint64_t> *tw;
    (void)hipMalloc(&data, bytes);
    (void)hipMalloc(&tw, 16384 * sizeof(TwPair<uint64_t>));
    (void)hipMalloc(&g_stamp, STAMP_WAVES * 16);
    (void)hipMemcpyToSymbol(HIP_SYMBOL(lab_buf), &g_stamp, sizeof g_stamp);
    std::vector<uint64_t> h(bytes / 8);
    uint64_t s = 88172645463325252ull;
    for (auto &v : h) {
        s ^= s << 13; s ^= s >> 7; s ^= s << 17;
        v = s % p;
    }
    (void)hipMemcpy(data, h.data(), bytes, hipMemcpyHostToDevice);
    std::vector<TwPair<uint64_t>> ht(16384);
    for (auto &t : ht) {
        s ^= s << 13; s ^= s >> 7; s ^= s << 17;
        t.w = s % p;
        t.ws = (uint64_t)((((unsigned __int128)t.w) << 64) / p);
    }
    (void)hipMemcpy(tw, ht.data(), ht.size() * sizeof(TwPair<uint64_t>), hipMemcpyHostToDevice);
    ModParams<uint64_t> P{};
    P.p = p; P.neg_p = 0 - p; P.two_p = 2 * p; P.neg_two_p = 0 - 2 * p;
    run_one<14, false, 4, LAB_TWC_F>(data, tw, P, bytes, 1);
    run_one<14, true, 4, LAB_TWC_I>(data, tw, P, bytes, 1);
#ifndef LAB_ONLY14
    run_one<13, false, 4, LAB_TWC_F>(data, tw, P, bytes, 2);
    run_one<13, true, 4, LAB_TWC_I>(data, tw, P, bytes, 2);
    run_one<12, false, 4, LAB_TWC_F>(data, tw, P, bytes, 4);
    run_one<12, true, 4, LAB_TWC_I>(data, tw, P, bytes, 4);
#endif
    return 0;
}
