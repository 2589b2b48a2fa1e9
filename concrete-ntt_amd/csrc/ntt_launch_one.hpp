// One launch of the LDS-resident transform kernels for a fixed (T, LOGN, INV, CLS, SUB): persistent
// software-pipelined kernel where eligible, one polynomial per workgroup otherwise.  Shared by the integer-class
// instantiation units (ntt_inst.inc) and the CLS_FP unit (ntt_inst_u64_fp.hip).
#pragma once
#include "ntt_kernel.hpp"
#include "ntt_launch.hpp"

namespace cntt {

static constexpr int WP_BLOCK = 256, WP_BLOCKS_PER_CU = 3;

static int num_cus() {  // compute units of the current device (cached per device)
    static int cached[64] = {0};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 256;
    if (cached[dev] == 0) {
        int n = 0;
        if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
        cached[dev] = n;
    }
    return cached[dev];
}

// sizes served by the persistent kernel: multi-pass, at most 256 threads per polynomial, and a thread-dependent
// twiddle image of at most 32 KiB (so that three workgroups fit the 160 KiB of LDS of a CU)
template <class K> static constexpr bool wp_eligible() {
    return K::NPASS > 1 && K::TPP <= WP_BLOCK && (size_t)K::IMG_ENTRIES * sizeof(TwPair<typename K::elem_t>) <= 32768;
}

// larger LDS-resident sizes (image too big): the same persistent walk with twiddles from L2, one polynomial per
// workgroup of N / E <= 1024 threads, as many workgroups per CU as LDS and 16 wavefronts (128 VGPRs) allow
template <class K> static constexpr bool wpg_eligible() { return K::NPASS > 1 && K::TPP >= 256 && K::TPP <= 1024 && K::LOGE <= 4; }

// Instances of ntt_kernel_wpg that hipcc (ROCm 7.2) cannot fit into the 128 VGPRs of four waves per SIMD without
// spilling.  A spilled register must never meet the asynchronous prefetch, so they are not instantiated at all;
// tests/test_host_plan.py (zero spills) and tests/test_async_load_guard.py fail if this list falls behind the compiler.
static constexpr bool wpg_spills(int logn, bool inv, int cls) {
    return inv && ((cls == CLS_PM64 && logn == 14) || (cls == CLS_STRICT && logn >= 13));
}

template <class T, int LOGN, bool INV, int CLS, bool SUB>
static hipError_t launch_one(T *data, const TwPair<T> *tw, const ModParams<T> &P, uint32_t nsub, uint32_t depth,
                             hipStream_t stream) {
    using K = NttKernel<T, LOGN, INV, CLS, SUB>;
    if (nsub == 0) return hipSuccess;
    if constexpr (wp_eligible<K>() && !SUB) {
        // polynomial inside one wavefront: persistent software-pipelined kernel, WP_BLOCKS_PER_CU
        // workgroups of 256 threads per CU walking tiles of PPB polynomials
        using W = NttWp<T, LOGN, INV, CLS, WP_BLOCK>;
        // 32 coefficients per thread need more than the 168 VGPRs that three waves per SIMD allow
        constexpr int BPC = (K::LOGE >= 5) ? 2 : WP_BLOCKS_PER_CU;
        const uint32_t ntiles = (nsub + W::PPB - 1) / W::PPB;
        uint32_t grid = (uint32_t)num_cus() * BPC;
        if (grid > ntiles) grid = ntiles;
        hipLaunchKernelGGL((ntt_kernel_wp<T, LOGN, INV, CLS, WP_BLOCK, BPC>), dim3(grid), dim3(WP_BLOCK), 0, stream, data,
                           tw, P, nsub);
    } else if constexpr (wpg_eligible<K>() && !SUB && sizeof(T) == 8 && CLS != CLS_GENERIC &&
                         !wpg_spills(LOGN, INV, CLS)) {
        // Measured (profiles/r02_bench_grid_table.txt vs r01_v6): the persistent walk pays for the 64-bit classes whose
        // butterflies are cheap enough to expose memory latency (CLS_FP +30..45 %, lazy / strict forward +5..10 %); the
        // 32-bit transforms and the Montgomery class ran 5..20 % slower on it and stay on ntt_kernel, and so do the
        // instances of wpg_spills().
        constexpr int WPB = K::TPP;
        using W = NttWp<T, LOGN, INV, CLS, WPB>;
        constexpr size_t LDS_BYTES = ((size_t)W::PPB << LOGN) * sizeof(T);
        constexpr int BY_LDS = (int)((160 * 1024) / LDS_BYTES), BY_WAVES = 16 / (WPB / 64);
        constexpr int BPC = BY_LDS < BY_WAVES ? BY_LDS : BY_WAVES;
        static_assert(BPC >= 1, "one polynomial must fit the LDS of a CU");
        const uint32_t ntiles = (nsub + W::PPB - 1) / W::PPB;
        uint32_t grid = (uint32_t)num_cus() * BPC;
        if (grid > ntiles) grid = ntiles;
        hipLaunchKernelGGL((ntt_kernel_wpg<T, LOGN, INV, CLS, WPB, 4>), dim3(grid), dim3(WPB), 0, stream, data, tw, P, nsub);
    } else {
        const uint32_t grid = (nsub + K::PPB - 1) / K::PPB;
        hipLaunchKernelGGL((ntt_kernel<T, LOGN, INV, CLS, SUB>), dim3(grid), dim3(K::BLOCK), 0, stream, data, tw, P,
                           nsub, depth);
    }
    return hipGetLastError();
}

}  // namespace cntt
