// Launch dispatch for the LDS-resident NTT kernels: (LOGN, class, SUB) -> template instance.
// The four instantiation units ntt_inst_{u64,u32}_{fwd,inv}.hip each define one launch_ntt<T, INV>.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "ntt_arith.hpp"

namespace cntt {
// The ONE switchboard behind cntt_debug_set() / cntt_debug_get() (include/cntt.h, "testing only"): kernel-selection overrides for A/B
// timing and for the parity tests that compare two device paths.  Results are identical either way; nothing in the library reads the
// process environment.  -1 = the library's own choice.  Defined in host.hip.
enum DebugSwitch : int {
    DBG_FP = 0,         // 1 (default): double-precision classes for u64 p < 2^51 / u32 p >= 2^31; 0: integer butterflies.  Read at plan creation.
    DBG_PM64,           // 1 (default): fold-by-c class for p = 2^64 - c; 0: Montgomery class.  Read at plan creation.
    DBG_BLK,            // 1 (default): wave-block walk for u64 N = 4096 ... 16384 / u32 N = 8192 ... 32768; 0: one polynomial per workgroup
    DBG_MUL32_BLK,      // 1 (default): fused product of 32-bit words at N = 16384 / 32768 on the wave-block walk
    DBG_EXT32_BLK,      // 1 (default): one-output mul_accumulate chain of 32-bit words on the wave-block walk
    DBG_EXT_ONE,        // 1 (default): fused chain kernel for 32-bit words above N = 4096; 0: composed pipeline
    DBG_EXT_SPLIT,      // -1 (default): split launches of 3 / 4 outputs where they measured faster; 0 never; 1 always
    DBG_NATIVE_ACC,     // 1 (default): accumulating-CRT whole-product kernels; 0: the parked-tile kernels
    DBG_PRODUCT_FUSED,  // -1 (default): product::Plan composed forward, fused inverse; 0 neither fused; 1 both fused
    DBG_PLAN52_VIA32,   // 1 (default): negacyclic_polymul of the Plan52 native kinds runs the Plan32 whole-product kernel; 0: composed on 50-bit primes
    DBG_COUNT
};
int debug_switch(DebugSwitch key);
}  // namespace cntt

namespace cntt {

// largest transform one workgroup keeps in LDS (gen_sched.py MAX_LDS_BYTES = 128 KiB)
template <class T> struct MaxLdsLogN;
template <> struct MaxLdsLogN<uint64_t> { static constexpr int value = 14; };
template <> struct MaxLdsLogN<uint32_t> { static constexpr int value = 15; };
// largest size whose transforms run in CLS_FPW: the 32768-point kernel (one 1024-thread workgroup, 128 VGPRs) would spill
constexpr int MAX_FPW_LOGN = 15;
template <class T> struct MinLogN;
template <> struct MinLogN<uint64_t> { static constexpr int value = 4; };  // src/prime64.rs:709
template <> struct MinLogN<uint32_t> { static constexpr int value = 5; };  // src/prime32.rs:635

// Runs nsub independent 2^logn-point transforms stored back to back at `data`.
// depth > 0: each transform is sub-block (sub mod 2^depth) of a 2^(logn+depth)-point polynomial
// whose top `depth` stages are done by global_stage_kernel (logn must be MaxLdsLogN then).
template <class T, bool INV>
hipError_t launch_ntt(int logn, int cls, T *data, const TwPair<T> *tw, const ModParams<T> &P, uint32_t nsub,
                      uint32_t depth, hipStream_t stream);

// The 64-bit-only classes CLS_FP / CLS_FP51 (p < 2^50 / 2^51) and CLS_PM64 (p = 2^64 - c): instances of the two launches
// above, each class in a translation unit of its own (ntt_inst_u64_fp.hip, ntt_inst_u64_fp51.hip, ntt_inst_u64_pm.hip).
// `tw` / `twf` / `twi` are the plan's tables for that class ((c, c/p) doubles; plain residues).
template <int CLS>
hipError_t launch_ntt_fp(int logn, bool inv, uint64_t *data, const TwPair<uint64_t> *tw, const ModParams<uint64_t> &P,
                         uint32_t nsub, hipStream_t stream);
template <int CLS>
hipError_t launch_mul_ntt_fp(int logn, uint64_t *lhs, const uint64_t *rhs_ntt, const TwPair<uint64_t> *twf,
                             const TwPair<uint64_t> *twi, const ModParams<uint64_t> &P, uint32_t nsub, hipStream_t stream);

// Fused lhs <- inv(mul_assign_normalize(fwd(lhs), rhs_ntt)) for transforms that live in one wavefront
// (2^logn <= 1024).  Returns hipErrorNotSupported for other sizes (the caller then runs three launches).
template <class T>
hipError_t launch_mul_ntt(int logn, int cls, T *lhs, const T *rhs_ntt, const TwPair<T> *twf, const TwPair<T> *twi,
                          const ModParams<T> &P, uint32_t nsub, hipStream_t stream);

// Fused mul_accumulate chain (ExtWp): out[b][o] (+)= inv(sum_j fwd(terms[b][j]) . key_ntt[j][o]), o < NOUT <= 4.
// Same eligibility as launch_mul_ntt; hipErrorNotSupported otherwise (the caller composes batched launches).
// One explicit specialisation per (T, NOUT), each in its own translation unit (ntt_inst_*_ext<NOUT>.hip).
template <class T, int NOUT>
hipError_t launch_ext_ntt_n(int logn, int cls, T *out, const T *terms, const T *key_ntt, const TwPair<T> *twf,
                            const TwPair<T> *twi, const ModParams<T> &P, uint32_t batch, uint32_t nterms, bool accumulate,
                            hipStream_t stream, uint32_t ostride);
bool ext_split_enabled();   // switch "ext_split" = 0: no split launches (A/B timing: the composed path instead)
template <class T>
inline hipError_t launch_ext_ntt(int logn, int cls, T *out, const T *terms, const T *key_ntt, const TwPair<T> *twf,
                                 const TwPair<T> *twi, const ModParams<T> &P, uint32_t batch, uint32_t nterms, uint32_t nout,
                                 bool accumulate, hipStream_t stream) {
    hipError_t e = hipErrorNotSupported;
    switch (nout) {
    case 0: return hipSuccess;
    case 1: e = launch_ext_ntt_n<T, 1>(logn, cls, out, terms, key_ntt, twf, twi, P, batch, nterms, accumulate, stream, 1); break;
    case 2: e = launch_ext_ntt_n<T, 2>(logn, cls, out, terms, key_ntt, twf, twi, P, batch, nterms, accumulate, stream, 2); break;
    case 3: e = launch_ext_ntt_n<T, 3>(logn, cls, out, terms, key_ntt, twf, twi, P, batch, nterms, accumulate, stream, 3); break;
    case 4: e = launch_ext_ntt_n<T, 4>(logn, cls, out, terms, key_ntt, twf, twi, P, batch, nterms, accumulate, stream, 4); break;
    default: return hipErrorNotSupported;
    }
    if (e != hipErrorNotSupported || nout < 3 || !ext_split_enabled()) return e;
    // Three / four outputs where only the one- / two-output kernels exist (64-bit words at n = 16384, 32-bit words at n = 32768:
    // the accumulator tiles do not fit 128 VGPRs): TWO fused launches of <= 2 outputs.  The terms are transformed twice, but
    // nothing else touches HBM; the composed path moves (2J + 3JO + 2O) n words in five launches.
    (void)hipGetLastError();
    e = launch_ext_ntt_n<T, 2>(logn, cls, out, terms, key_ntt, twf, twi, P, batch, nterms, accumulate, stream, nout);
    if (e != hipSuccess) return e;
    T *out2 = out + ((size_t)2 << logn);
    const T *key2 = key_ntt + ((size_t)2 << logn);
    if (nout == 3) return launch_ext_ntt_n<T, 1>(logn, cls, out2, terms, key2, twf, twi, P, batch, nterms, accumulate, stream, nout);
    return launch_ext_ntt_n<T, 2>(logn, cls, out2, terms, key2, twf, twi, P, batch, nterms, accumulate, stream, nout);
}

// Whole negacyclic_polymul of one native Plan32 kind (native_fused.hpp) for 32 <= n <= 16384; hipErrorNotSupported
// otherwise.  `tables` points to the FusedTables<KP> of the plan (native_fused.hpp); KIND = cntt_native_kind_t value.
struct SplitArgs;
struct CrtArgs;
struct AccArgs;
struct ProductArgs;
// product::Plan with two u32 primes of one arithmetic class `cls`, 32 <= n <= 4096 (product_fused.hpp): forward
// (flag = FwdMode::Bounded applies) or inverse (flag = InvMode::Accumulate) in one kernel; `tables` points to a
// ProductFusedTables.  hipErrorNotSupported for other sizes.
hipError_t launch_product_fused2(int logn, int cls, bool inv, uint64_t *standard, uint32_t *res32, const void *tables,
                                 const ProductArgs &A, uint32_t batch, bool flag, hipStream_t st);
// The persistent form of the kernel (n = 16384 / 32768 of every kind, n = 8192 of the kinds that do not fit LDS there, native128 =
// kind 2 at every size) parks residue
// tiles in `scratch` (native_fused_scratch_words() 32-bit words); the other shapes ignore it.
// `acc` (round 4): constants of the accumulating CRT and `tables_acc`, the FusedTables whose last-stage constants carry
// (M / P_i)^-1 / n -- the sizes that have the register-resident kernel (native_fused_acc()) run it unless the testing switch native_acc is 0.
template <int KIND>
hipError_t launch_native_fused(int logn, void *prod, const void *lhs, const void *rhs, const void *tables, const SplitArgs &S,
                               const CrtArgs &C, uint32_t batch, uint32_t *scratch, hipStream_t st, const AccArgs *acc,
                               const void *tables_acc);
// sizes / kinds of native_polymul_kernel_acc (no workspace, no parking)
// (words of at most 64 bits up to n = 16384 -- 16 coefficients per thread, three words of state each; the 16-byte words of
// native128 / native_binary128 -- five words of state -- up to n = 4096, where a product is at most 256 threads at 168 VGPRs, and, round 5,
// native128 at n = 8192: 128 VGPRs with 15 spilled registers, 887 -> 763 ns per product; measured SLOWER than the parked-tile kernels and
// therefore not enabled: native128 n = 16384 1990 against 1920 ns, native_binary128 n = 8192 433 against 387, n = 16384 1030 against 838
// -- profiles/r05_native128_acc_ab.txt)
constexpr bool native_fused_acc(int kind, int logn) {
    return kind == 2 ? (logn >= 5 && logn <= 13) : kind == 5 ? (logn >= 5 && logn <= 12) : (kind >= 0 && kind <= 4 && logn >= 5 && logn <= 14);
}
bool native_acc_enabled();
inline int device_num_cus() {
    int dev = 0, n = 0;
    if (hipGetDevice(&dev) != hipSuccess) return 256;
    if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
    return n;
}
constexpr bool native_fused_lds13(int kind) { return kind == 0 || kind == 1 || kind == 4; }  // n = 8192 with the tiles in LDS (native_fused_inst.inc)
inline bool native_fused_persistent(int kind, int logn) {
    if (native_fused_acc(kind, logn) && native_acc_enabled()) return false;
    return (logn == 13 && !native_fused_lds13(kind)) || logn == 14 || logn == 15 || (kind == 2 && logn >= 5 && logn <= 12);
}
// words of one residue tile of a workgroup (256 threads x 16 coefficients hold 4096 / n products below n = 4096)
inline size_t native_fused_tile_words(int logn) { return logn < 12 ? (size_t)4096 : (size_t)1 << logn; }
// workgroups of the persistent kernel: what is resident on `ncu` compute units (four 256-thread workgroups per unit for
// native128 below n = 8192, two 512-thread ones at n = 8192, one 1024-thread one at n = 16384 / 32768), at most one per group of products
inline uint32_t native_fused_grid(int logn, int ncu, uint32_t batch) {
    const size_t ppb = native_fused_tile_words(logn) >> logn;
    const size_t groups = ((size_t)batch + ppb - 1) / ppb, g = (size_t)ncu * (logn >= 14 ? 1u : logn == 13 ? 2u : 4u);
    return (uint32_t)(g < groups ? g : groups);
}
inline size_t native_fused_scratch_words(int kind, int logn, int nprimes, int ncu, uint32_t batch) {
    if (!native_fused_persistent(kind, logn)) return 0;
    return (size_t)native_fused_grid(logn, ncu, batch) * (size_t)(nprimes - 1) * native_fused_tile_words(logn);
}

}  // namespace cntt
