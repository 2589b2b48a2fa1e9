// Census of a fold-class forward butterfly for p = 2^62 - c (c < 2^22) next to the repo's Shoup butterfly.
// Build + count: hipcc -O3 -std=c++17 --offload-arch=gfx950 --cuda-device-only -Iconcrete-ntt_amd/csrc -S tools/fold_census.hip -o /tmp/fold_census.s, then count the v_* lines between the BEGIN_ / END_ markers (profiles/r04_fold_class_census.txt).
#include "ntt_arith.hpp"
using namespace cntt;
typedef unsigned __int128 u128;
__device__ __forceinline__ uint64_t fold_mul(uint64_t y, uint64_t w, uint32_t c) {
    const u128 z = (u128)y * w;                               // < 2^126
    const uint64_t A = (uint64_t)(z >> 62), B = (uint64_t)z & 0x3fffffffffffffffull;
    const u128 u = (u128)A * c;                               // < 2^86
    const uint64_t A1 = (uint64_t)(u >> 62), B1 = (uint64_t)u & 0x3fffffffffffffffull;
    return B + B1 + A1 * c;                                   // < 2^63 + 2^46
}
extern "C" __global__ void k_fold(uint64_t *d, const uint64_t *tw, uint32_t c, uint64_t two_p, uint64_t neg_two_p) {
    uint64_t x = d[threadIdx.x], y = d[threadIdx.x + 64];
    const uint64_t w = tw[threadIdx.x];
    asm volatile("; BEGIN_FOLD" ::: "memory");
    x = csub_two_p<uint64_t>(x, two_p, neg_two_p);
    const uint64_t t = fold_mul(y, w, c);
    const uint64_t X = x + t, Y = x + two_p - t;
    asm volatile("; END_FOLD" :: "v"(X), "v"(Y) : "memory");
    d[threadIdx.x] = X; d[threadIdx.x + 64] = Y;
}
extern "C" __global__ void k_shoup(uint64_t *d, const TwPair<uint64_t> *tw, ModParams<uint64_t> P) {
    uint64_t x = d[threadIdx.x], y = d[threadIdx.x + 64];
    const TwPair<uint64_t> w = tw[threadIdx.x];
    asm volatile("; BEGIN_SHOUP" ::: "memory");
    Bfly<uint64_t, CLS_LAZY>::template fwd<false, false>(x, y, w.w, w.ws, P);
    asm volatile("; END_SHOUP" :: "v"(x), "v"(y) : "memory");
    d[threadIdx.x] = x; d[threadIdx.x + 64] = y;
}
