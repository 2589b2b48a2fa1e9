// One launch of the fused product kernel (MulWp) for a fixed (T, LOGN, CLS) and its workgroup shape.  Shared by the
// integer-class instantiation units (ntt_mul_inst.inc) and the CLS_FP unit (ntt_inst_u64_fp.hip).
#pragma once
#include <cstdlib>
#include "ntt_blk.hpp"
#include "ntt_kernel.hpp"
#include "ntt_launch.hpp"

namespace cntt {

// Workgroup shape of the fused product kernel: as many wavefronts per CU as LDS (160 KiB: one exchange buffer per
// polynomial in flight + the forward and inverse twiddle images, shared by the workgroup) and registers allow,
// up to 12 (three per SIMD, 168 VGPRs).  256-thread workgroups when three of them fit; otherwise ONE 768-thread
// workgroup per CU, which pays for the images once (16 coefficients per thread only: the 32-coefficient u32
// kernels need more than 168 registers); otherwise 256 x 2 or 512 x 1.
template <class T, int LOGN> struct MulShape {
    using K0 = NttKernel<T, LOGN, false, CLS_LAZY, false>;
    static constexpr size_t IMG = (size_t)K0::IMG_ENTRIES * sizeof(TwPair<T>);
    static constexpr size_t LDS_MAX = 160 * 1024;
    static constexpr size_t lds(int block) { return (size_t)(block / K0::TPP) * K0::LDS_WORDS_1 * sizeof(T) + 2 * IMG; }
    static constexpr bool THREE_SMALL = 3 * lds(256) <= LDS_MAX && K0::E == 16;
    static constexpr bool ONE_BIG = !THREE_SMALL && lds(768) <= LDS_MAX && K0::E == 16 && K0::TPP <= 256 && 768 % K0::TPP == 0;
    static constexpr bool TWO_SMALL = 2 * lds(256) <= LDS_MAX;
    static constexpr int BLOCK = THREE_SMALL ? 256 : ONE_BIG ? 768 : TWO_SMALL ? 256 : 512;
    static constexpr int PER_CU = THREE_SMALL ? 3 : ONE_BIG ? 1 : TWO_SMALL ? 2 : 1;
    static constexpr int WAVES_PER_SIMD = BLOCK / 64 * PER_CU / 4;  // __launch_bounds__ second argument (HIP: per EU)
    static_assert(lds(BLOCK) * PER_CU <= LDS_MAX, "fused product kernel does not fit LDS");
};

static int mul_num_cus() {
    int dev = 0, n = 0;
    if (hipGetDevice(&dev) != hipSuccess) return 256;
    if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
    return n;
}

template <class T, int LOGN, int CLS>
static hipError_t mul_one(T *lhs, const T *rhs, const TwPair<T> *twf, const TwPair<T> *twi, const ModParams<T> &P,
                          uint32_t nsub, hipStream_t stream) {
    using SH = MulShape<T, LOGN>;
    using K = MulWp<T, LOGN, CLS, SH::BLOCK>;
    const uint32_t ntiles = (nsub + K::PPB - 1) / K::PPB;
    uint32_t grid = (uint32_t)mul_num_cus() * SH::PER_CU;
    if (grid > ntiles) grid = ntiles;
    hipLaunchKernelGGL((mul_kernel_wp<T, LOGN, CLS, SH::BLOCK, SH::WAVES_PER_SIMD>), dim3(grid), dim3(SH::BLOCK), 0, stream,
                       lhs, rhs, twf, twi, P, nsub);
    return hipGetLastError();
}

// Fused product on the wave-block walk (MulBlk): 64-bit words, N = 4096 ... 16384, every class but the Montgomery one.
constexpr bool mul_blk_eligible(int bytes, int logn, int cls) { return bytes == 8 && logn >= 12 && logn <= 14 && cls != CLS_GENERIC; }
// Workgroup shape: N/16 threads, four waves per SIMD (116-128 VGPRs with the register prefetch of the next polynomial --
// it fits once the inverse half recomputes its addresses from a fresh opaque copy of the thread index instead of keeping
// the forward half's alive), as many workgroups per CU as LDS and sixteen wavefronts allow: 4 / 2 / 1.
// The CPU tests read the code objects: zero spills in every instance.
template <int LOGN, int CLS> struct MulBlkShape {
    static constexpr int WPW = 4;
    static constexpr int PER_CU = LOGN == 12 ? 4 : LOGN == 13 ? 2 : 1;
    static constexpr bool PREFETCH = true;
    static constexpr int TWC = 2;   // (one pair until late round 4: 110 ... 118 VGPRs with two, no spills; -1 ... -3.6 % in ten of twelve shapes, +1 % in two)
};
template <class T, int LOGN, int CLS>
static hipError_t mul_blk_one(T *lhs, const T *rhs, const TwPair<T> *twf, const TwPair<T> *twi, const ModParams<T> &P,
                              uint32_t nsub, hipStream_t stream) {
    using SH = MulBlkShape<LOGN, CLS>;
    using K = MulBlk<T, LOGN, CLS, SH::TWC, SH::PREFETCH>;
    uint32_t grid = (uint32_t)mul_num_cus() * SH::PER_CU;
    if (grid > nsub) grid = nsub;
    hipLaunchKernelGGL((mul_kernel_blk<T, LOGN, CLS, SH::WPW, SH::TWC, SH::PREFETCH>), dim3(grid), dim3(K::WPB), 0, stream, lhs, rhs,
                       twf, twi, P, nsub);
    return hipGetLastError();
}

// 32-bit words, N = 16384 / 32768 on the same walk (2048-word blocks, 32 coefficients per thread; two 512-thread / one 1024-thread workgroup per CU)
// where it beats the one-polynomial-per-workgroup kernel (MulOne); ordinary loads except for the one shape with the registers to spare (an
// asynchronous load must not meet a spilled register).  Same box, ns per product, walk / MulOne (tools/mul_bench.py, testing switch mul32_blk = 0 for the latter):
//   30-bit  n = 16384  71.3 / 73.8 (-3.4 %, with the prefetch; -1.7 % without)   n = 32768  156.5 / 173.5 (-9.8 %)
//   31-bit  n = 16384  81.5 / 75.4 (+8.1 %: not used)                             n = 32768  176.5 / 195.7 (-9.8 %)
//   p>=2^31 n = 16384  95.2 / 101.4 (-6.2 %)                                      n = 32768  216.2 / 204.1 (+5.9 %, 17 spilled registers: not used)
constexpr bool mul32_blk_wins(int logn, int cls) {
    return (cls == CLS_LAZY && (logn == 14 || logn == 15)) || (cls == CLS_STRICT && logn == 15) || (cls == CLS_FPW && logn == 14);
}
static bool mul32_blk_enabled() { return debug_switch(DBG_MUL32_BLK) != 0; }
template <class T, int LOGN, int CLS>
static hipError_t mul32_blk_one(T *lhs, const T *rhs, const TwPair<T> *twf, const TwPair<T> *twi, const ModParams<T> &P,
                                uint32_t nsub, hipStream_t stream) {
    constexpr bool PF = LOGN == 14 && CLS == CLS_LAZY;   // 112 VGPRs with the prefetch in flight, no spills
    using K = MulBlk<T, LOGN, CLS, 2, PF>;
    uint32_t grid = (uint32_t)mul_num_cus() * (LOGN == 14 ? 2 : 1);
    if (grid > nsub) grid = nsub;
    hipLaunchKernelGGL((mul_kernel_blk<T, LOGN, CLS, 4, 2, PF>), dim3(grid), dim3(K::WPB), 0, stream, lhs, rhs, twf, twi, P, nsub);
    return hipGetLastError();
}

// N = 32768, 64-bit words: Mul32k (ntt_blk.hpp), one 1024-thread workgroup per CU
template <class T, int CLS>
static hipError_t mul_32k_one(T *lhs, const T *rhs, const TwPair<T> *twf, const TwPair<T> *twi, const ModParams<T> &P,
                              uint32_t nsub, hipStream_t stream) {
    uint32_t grid = (uint32_t)mul_num_cus();
    if (grid > nsub) grid = nsub;
    hipLaunchKernelGGL((mul_kernel_32k<T, CLS, 4>), dim3(grid), dim3(Mul32k<T, CLS>::WPB), 0, stream, lhs, rhs, twf, twi, P, nsub);
    return hipGetLastError();
}

}  // namespace cntt
