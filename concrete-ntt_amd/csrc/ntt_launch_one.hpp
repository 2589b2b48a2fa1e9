// One launch of the LDS-resident transform kernels for a fixed (T, LOGN, INV, CLS, SUB): persistent
// software-pipelined kernel where eligible, one polynomial per workgroup otherwise.  Shared by the integer-class
// instantiation units (ntt_inst.inc) and the CLS_FP unit (ntt_inst_u64_fp.hip).
#pragma once
#include <cstdlib>

#include "ntt_blk.hpp"
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

// Wave-block kernel (ntt_blk.hpp): 64-bit words, N = 4096 ... 16384, every class but the Montgomery one.
// BLK_TWC: twiddle pairs a thread loads at a time in the block passes -- what fits 128 VGPRs without spilling
// (tests/test_host_plan.py and tests/test_async_load_guard.py check the code objects).
// (The inverse of the strict class at N >= 8192 spilled at 128 VGPRs until the padded exchange layout freed the swizzle's
// address registers: 112 VGPRs now, instantiated like the rest.)
// Round 4: 32-bit words, N = 16384 / 32768 (2048-word blocks, 32 coefficients per thread).
template <class T, int LOGN, bool INV, int CLS, bool SUB> static constexpr bool blk_eligible() {
    // Measured against the plain kernel on one box (profiles/r04_blk32_vs_plain.txt): n = 16384 +10 % / +6 % (30-bit fwd / inv), +14 % (31-bit
    // fwd); n = 32768 +12 % / +32 %, +28 %; n = 8192 -1 ... -4 % -- four 512-thread workgroups per CU already overlap there: not used.
    // The strict class's inverse and the double-precision class for p >= 2^31 want ~170 registers at 32 coefficients per thread and spill
    // 20 ... 48 of them next to the asynchronous prefetch, which is fatal (tests/test_async_load_guard.py); WITHOUT the prefetch (ordinary
    // loads where the transform begins, NttBlk::run<false>: 102 ... 118 VGPRs, no spills) the walk still wins where three or four workgroups per
    // CU do not already overlap: p >= 2^31 n = 32768 141.0 -> 102.6 / 127.7 -> 104.2 ns (23.3 -> 31.9 % / 25.7 -> 31.5 % of the roofline), n = 16384
    // 56.0 -> 49.4 / 57.4 -> 48.1 ns (29.2 -> 33.2 % / 28.5 -> 34.1 %), n = 8192 +6 % / +1 % slower (not used); 31-bit inverse n = 32768 118.0 -> 97.8 ns
    // (27.8 -> 33.5 %), n = 16384 +-4 % over two boxes (not used).  profiles/r04_blk32_noprefetch_ab.txt
    if (sizeof(T) == 4) return !SUB && LOGN >= 14 && LOGN <= 15 && (CLS == CLS_LAZY || CLS == CLS_FPW || (CLS == CLS_STRICT && (!INV || LOGN == 15)));
    return sizeof(T) == 8 && !SUB && LOGN >= 12 && LOGN <= 14 && CLS != CLS_GENERIC;
}
// the shapes that spill run the walk without the asynchronous prefetch
template <class T, bool INV, int CLS> static constexpr bool blk_prefetch() {
    return !(sizeof(T) == 4 && (CLS == CLS_FPW || (CLS == CLS_STRICT && INV)));
}
// (double-buffered: 2 * TWC pairs are in flight; the forward kernels hold the prefetch across these passes, the inverse ones
// issue it behind them.  Until late round 4 the forward kernels took one pair at a time -- two fit since the padded exchange layout freed
// the swizzle's address registers (108 ... 118 VGPRs, no spills): 62-bit N = 16384 0.348 -> 0.362 of the roofline (C4's forward pass), N = 4096
// -2.4 %, 63-bit -2.4 ... -4.5 %, 2^64 - c -0.4 ... -2.7 %, the double-precision classes +-1 %; same-box A/B, profiles/r04_blk_twc_ab.txt)
// (the strict class's inverse too: 106 ... 108 VGPRs on the reference-form butterflies, -1.1 ... -1.9 %)
static constexpr int blk_twc(int logn, bool inv, int cls) { return 2; }
static bool blk_enabled() { return debug_switch(DBG_BLK) != 0; }   // (A/B runs: one polynomial per workgroup, no persistent walk)

template <class T, int LOGN, bool INV, int CLS, bool SUB>
static hipError_t launch_one(T *data, const TwPair<T> *tw, const ModParams<T> &P, uint32_t nsub, uint32_t depth,
                             hipStream_t stream) {
    using K = NttKernel<T, LOGN, INV, CLS, SUB>;
    if (nsub == 0) return hipSuccess;
    if constexpr (blk_eligible<T, LOGN, INV, CLS, SUB>()) {
        if (blk_enabled()) {
            using W = NttBlk<T, LOGN, INV, CLS>;
            constexpr size_t LDS_BYTES = sizeof(T) * W::B::LDS_WORDS_1;
            constexpr int BY_LDS = (int)((160 * 1024) / LDS_BYTES), BY_WAVES = 16 / (W::WPB / 64);
            constexpr int BPC = BY_LDS < BY_WAVES ? BY_LDS : BY_WAVES;
            uint32_t grid = (uint32_t)num_cus() * BPC;
            if (grid > nsub) grid = nsub;
            hipLaunchKernelGGL((ntt_kernel_blk<T, LOGN, INV, CLS, 4, (sizeof(T) == 4 ? 2 : blk_twc(LOGN, INV, CLS)), blk_prefetch<T, INV, CLS>()>),
                               dim3(grid), dim3(W::WPB), 0, stream, data, tw, P, nsub);
            return hipGetLastError();
        }
    }
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
    } else {
        const uint32_t grid = (nsub + K::PPB - 1) / K::PPB;
        hipLaunchKernelGGL((ntt_kernel<T, LOGN, INV, CLS, SUB, plain_fam<T, LOGN, SUB>()>), dim3(grid), dim3(K::BLOCK), 0, stream, data,
                           tw, P, nsub, depth);
    }
    return hipGetLastError();
}

}  // namespace cntt
