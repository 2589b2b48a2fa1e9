// Elementwise kernels around the transforms: pointwise products, residue split, CRT
// recombination, synthetic-input fill, and the strided global stages used when a polynomial does
// not fit one workgroup's LDS.  All are HBM-streaming kernels: 16-byte accesses per lane,
// grid-stride loops capped at a few waves per SIMD.
#pragma once
#include "ntt_arith.hpp"

namespace cntt {

// ---------------------------------------------------------------------------------------------
// K3/K4: pointwise kernels.  src/prime64.rs:534-584,690-699 ; src/prime32.rs:383-408,477-488,575-598
// (values equal the reference's Barrett-then-Shoup results: canonical a*b*N^-1, a*N^-1, acc+a*b)
// ---------------------------------------------------------------------------------------------
enum : int { PW_MUL_NORMALIZE = 0, PW_NORMALIZE = 1, PW_MUL_ACCUMULATE = 2, PW_ADD = 3 };

// STREAM: the operands are larger than the 256 MiB Infinity Cache and pass through once -- non-temporal loads and stores (round 5:
// -5 % on 512 MiB operands).  Operands that fit the cache keep the default policy: with the hint a 128 MiB batch that the previous kernel
// left in the cache is fetched from HBM again (+12 ... +17 %, profiles/r05_small_batch_ab.txt).  The launcher decides (host.hip).
template <class T, int OP, bool STREAM = false>
__global__ __launch_bounds__(256) void pointwise_kernel(T *__restrict__ a, const T *__restrict__ b,
                                                        const T *__restrict__ c, const ModParams<T> P, size_t count) {
    // OP == MUL_NORMALIZE: a <- a*b*ninv ; NORMALIZE: a <- a*ninv ; MUL_ACCUMULATE: a <- a + b*c ; ADD: a <- a + b
    constexpr int NV = 16 / sizeof(T);
    using V = __attribute__((ext_vector_type(NV))) T;
    const bool generic = P.cls == CLS_GENERIC;
    const size_t nvec = count / NV;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    auto apply = [&](V &va, const V &vb, const V &vc) {
#pragma unroll
        for (int k = 0; k < NV; ++k) {
            if constexpr (OP == PW_MUL_NORMALIZE) va[k] = mul_normalize<T>(va[k], vb[k], P, generic);
            if constexpr (OP == PW_NORMALIZE) va[k] = normalize1<T>(va[k], P, generic);
            if constexpr (OP == PW_MUL_ACCUMULATE) va[k] = mul_acc<T>(va[k], vb[k], vc[k], P, generic);
            if constexpr (OP == PW_ADD) va[k] = add_mod<T>(va[k], vb[k], P.p);
        }
    };
    // two vectors per thread in flight (the loads of the second are issued before the arithmetic of the first)
    auto ld = [](const T *base, size_t i) -> V {
        if constexpr (STREAM) return __builtin_nontemporal_load(reinterpret_cast<const V *>(base) + i);
        else return reinterpret_cast<const V *>(base)[i];
    };
    auto st = [](T *base, size_t i, const V &v) {
        if constexpr (STREAM) __builtin_nontemporal_store(v, reinterpret_cast<V *>(base) + i);
        else reinterpret_cast<V *>(base)[i] = v;
    };
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (; i + stride < nvec; i += 2 * stride) {
        V va0 = ld(a, i), va1 = ld(a, i + stride);
        V vb0 = va0, vb1 = va1, vc0 = va0, vc1 = va1;
        if constexpr (OP != PW_NORMALIZE) {
            vb0 = ld(b, i);
            vb1 = ld(b, i + stride);
        }
        if constexpr (OP == PW_MUL_ACCUMULATE) {
            vc0 = ld(c, i);
            vc1 = ld(c, i + stride);
        }
        apply(va0, vb0, vc0);
        apply(va1, vb1, vc1);
        st(a, i, va0);
        st(a, i + stride, va1);
    }
    for (; i < nvec; i += stride) {
        V va = ld(a, i);
        V vb = va, vc = va;
        if constexpr (OP != PW_NORMALIZE) vb = ld(b, i);
        if constexpr (OP == PW_MUL_ACCUMULATE) vc = ld(c, i);
        apply(va, vb, vc);
        st(a, i, va);
    }
    // tail (count not a multiple of the vector width)
    for (size_t i = nvec * NV + (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += stride) {
        if constexpr (OP == PW_MUL_NORMALIZE) a[i] = mul_normalize<T>(a[i], b[i], P, generic);
        if constexpr (OP == PW_NORMALIZE) a[i] = normalize1<T>(a[i], P, generic);
        if constexpr (OP == PW_MUL_ACCUMULATE) a[i] = mul_acc<T>(a[i], b[i], c[i], P, generic);
        if constexpr (OP == PW_ADD) a[i] = add_mod<T>(a[i], b[i], P.p);
    }
}

// ---------------------------------------------------------------------------------------------
// mul_accumulate chain in the NTT domain, any transform size (the composed path behind
// cntt_prime*_external_product_batch when the fused ExtWp kernel does not cover the size):
//     acc[b][o][e] = sum_j t[b][j][e] * key[j][o][e] mod p        (src/prime64.rs:1085-1128 applied J times)
// one thread per 16-byte vector of one (b, o) output polynomial; key is shared by the batch (L2).
// ---------------------------------------------------------------------------------------------
template <class T>
__global__ __launch_bounds__(256) void ext_accumulate_kernel(T *__restrict__ acc, const T *__restrict__ t,
                                                             const T *__restrict__ key, const ModParams<T> P,
                                                             uint32_t logn, uint32_t nterms, uint32_t nout, size_t batch) {
    constexpr int NV = 16 / sizeof(T);
    using V = __attribute__((ext_vector_type(NV))) T;
    const bool generic = P.cls == CLS_GENERIC;
    const uint32_t lv = logn - (NV == 2 ? 1 : 2);  // log2 vectors per polynomial (n >= 16)
    const size_t total = (batch * nout) << lv;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
        const size_t v = i & (((size_t)1 << lv) - 1), bo = i >> lv;
        const size_t b = bo / nout, o = bo - b * nout;
        V a;
#pragma unroll
        for (int k = 0; k < NV; ++k) a[k] = 0;
        for (uint32_t j = 0; j < nterms; ++j) {
            const V x = reinterpret_cast<const V *>(t)[((b * nterms + j) << lv) + v];
            const V y = reinterpret_cast<const V *>(key)[(((size_t)j * nout + o) << lv) + v];
#pragma unroll
            for (int k = 0; k < NV; ++k) a[k] = mul_acc<T>(a[k], x[k], y[k], P, generic);
        }
        reinterpret_cast<V *>(acc)[i] = a;
    }
}

// ---------------------------------------------------------------------------------------------
// synthetic inputs: element i = mulhi(splitmix64(seed + i), bound)   (bound == 0: raw)
// (same definition as oracle/cntt_oracle.c orc_fill_uniform_*)
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ uint64_t splitmix64(uint64_t x) {
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}
template <class T> __global__ __launch_bounds__(256) void fill_uniform_kernel(T *dst, size_t count, T bound, uint64_t seed) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += stride) {
        const uint64_t r = splitmix64(seed + i);
        if constexpr (sizeof(T) == 8)
            dst[i] = bound ? mulhi((uint64_t)r, (uint64_t)bound) : r;
        else
            dst[i] = bound ? (uint32_t)(((r >> 32) * (uint64_t)bound) >> 32) : (uint32_t)(r >> 32);
    }
}

// ---------------------------------------------------------------------------------------------
// strided global stage: one radix-2 stage over whole polynomials in HBM, used for the top
// `depth` stages of transforms larger than one workgroup's LDS (the reference's depth-first
// recursion step, src/prime64/shoup.rs:660-682 / :1444-1466).
// ---------------------------------------------------------------------------------------------
template <class T, bool INV, int CLS>
__global__ __launch_bounds__(256) void global_stage_kernel(T *__restrict__ data, const TwPair<T> *__restrict__ tw,
                                                           const ModParams<T> P, uint32_t logn, uint32_t s,
                                                           size_t nbfly_total, bool finish) {
    // butterfly i of polynomial q: block = i / t, j = i % t, t = N >> (s+1)
    const uint32_t logt = logn - 1 - s;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t gi = (size_t)blockIdx.x * blockDim.x + threadIdx.x; gi < nbfly_total; gi += stride) {
        const size_t q = gi >> (logn - 1);
        const uint32_t i = (uint32_t)(gi & (((size_t)1 << (logn - 1)) - 1));
        const uint32_t blk = i >> logt, j = i & ((1u << logt) - 1u);
        T *x = data + (q << logn) + ((size_t)blk << (logt + 1)) + j;
        T *y = x + ((size_t)1 << logt);
        const TwPair<T> w = tw[(1u << s) + blk];
        T a = *x, b = *y;
        if constexpr (INV) {
            Bfly<T, CLS>::inv(a, b, w.w, w.ws, P);
            if (finish) {
                a = Bfly<T, CLS>::finish_inv(a, P);
                b = Bfly<T, CLS>::finish_inv(b, P);
            }
        } else {
            Bfly<T, CLS>::fwd(a, b, w.w, w.ws, P);
        }
        *x = a;
        *y = b;
    }
}

// ---------------------------------------------------------------------------------------------
// K5: residue split.  value % P_i for each prime (src/native64.rs:980-993 etc.), or plain
// truncation for the binary operand (src/native_binary64.rs:379-385).
// Layout: value[batch*N] words; residues[k] -> batch*N elements of R.
// Division-free: for the 30-bit primes a word is folded 32 bits at a time with Shoup products by
// c = 2^32 mod p (any 32-bit operand is allowed, results in [0,2p), sums stay below 4p < 2^32); the
// 50-bit primes use one Barrett step with floor(2^64 / p).  Results are canonical, i.e. equal to `%`.
// ---------------------------------------------------------------------------------------------
struct SplitArgs {
    void *res[10];
    uint64_t prime[10];
    uint32_t c[10], c_shoup[10];  // 2^32 mod p and floor(c * 2^32 / p)          (30-bit primes)
    uint32_t one_shoup[10];       // floor(2^32 / p): Shoup companion of 1        (30-bit primes)
    uint64_t barrett[10];         // floor(2^64 / p)                              (50-bit primes)
    int k;
};

// x mod p into [0, 2p) for any 32-bit x
__device__ __forceinline__ uint32_t red32_lazy(uint32_t x, uint32_t p, uint32_t one_shoup) {
    return x - __umulhi(x, one_shoup) * p;
}
// acc * 2^32 + limb  (mod p), lazily: acc < 2^32 arbitrary, result < 4p
__device__ __forceinline__ uint32_t fold32(uint32_t acc, uint32_t limb, uint32_t p, uint32_t c, uint32_t c_shoup,
                                           uint32_t one_shoup) {
    const uint32_t t = acc * c - __umulhi(acc, c_shoup) * p;  // [0, 2p)
    return t + red32_lazy(limb, p, one_shoup);                // [0, 4p)
}
__device__ __forceinline__ uint32_t canon4(uint32_t x, uint32_t p) {  // [0,4p) -> [0,p)
    x = umin<uint32_t>(x, x - 2 * p);
    return umin<uint32_t>(x, x - p);
}

template <class W, class R, bool BINARY>
__global__ __launch_bounds__(256) void split_kernel(const W *__restrict__ value, SplitArgs A, size_t count) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += stride) {
        uint32_t limb[4] = {0, 0, 0, 0};
        constexpr int NL = sizeof(W) / 4;
#pragma unroll
        for (int l = 0; l < NL; ++l) limb[l] = reinterpret_cast<const uint32_t *>(value)[i * NL + l];
        const uint64_t lo64 = (uint64_t)limb[0] | ((uint64_t)limb[1] << 32);
#pragma unroll 1
        for (int k = 0; k < A.k; ++k) {
            R r;
            if constexpr (BINARY) {
                r = (R)lo64;  // `*value as u32` / plain copy: src/native_binary64.rs:379-385, :481-487
            } else if constexpr (sizeof(R) == 8) {
                // 50-bit prime, u64 word (a u32 word never reaches here: it is copied, src/native32.rs:447-452)
                const uint64_t p = A.prime[k];
                const uint64_t q = mulhi(lo64, A.barrett[k]);
                uint64_t t = lo64 - q * p;  // [0, 2p)
                r = t >= p ? t - p : t;
            } else {
                const uint32_t p = (uint32_t)A.prime[k];
                uint32_t acc = limb[NL - 1];
                if constexpr (NL == 1) {
                    acc = red32_lazy(acc, p, A.one_shoup[k]);
                } else {
#pragma unroll
                    for (int l = NL - 2; l >= 0; --l) acc = fold32(acc, limb[l], p, A.c[k], A.c_shoup[k], A.one_shoup[k]);
                }
                r = canon4(acc, p);
            }
            reinterpret_cast<R *>(A.res[k])[i] = r;
        }
    }
}

// ---------------------------------------------------------------------------------------------
// K6: CRT recombination (mixed radix / Garner, centred lift decided by the TOP digit, wrapping)
//   src/native32.rs:28-56, src/native64.rs:91-141, src/native128.rs:20-118,
//   src/native_binary32.rs:22-41, src/native_binary64.rs:33-61, src/native_binary128.rs:13-63,
//   52-bit: src/native32.rs:223-253, src/native64.rs:770-829, src/native_binary32.rs:111-125,
//           src/native_binary64.rs:230-260
// The digit recurrence is generic over "groups": digit g has modulus M[g] (one prime or a product
// of two primes, < 2^62), residue r_g (combined inside a pair group first), and
//   v_g = (r_g - (v_0 + M_0 (v_1 + M_1 (...)))) * inv_g  mod M_g ,  inv_g = (M_0 ... M_{g-1})^-1 mod M_g
// The digits are unique in [0, M_g), so they equal the reference's v0, v12, v34, ... whatever
// product formula computes them.  All constants are computed on the host (exact) and passed by value.
// ---------------------------------------------------------------------------------------------
struct CrtArgs {
    const void *res[10];
    int k;              // residue arrays
    int ngroups;        // number of digits
    int ga[5], gb[5];   // prime indices of each group; gb < 0: the group is the single prime ga
    uint64_t prime[10];
    uint64_t pair_inv[5];        // pair groups: P_a^-1 mod P_b
    uint32_t pair_inv_shoup[5];  // floor(pair_inv * 2^32 / P_b)
    uint64_t M[5];               // group moduli (a prime or a product of two 30-bit primes, < 2^62)
    uint64_t inv[5];             // inv[g] = (M_0..M_{g-1})^-1 mod M[g]  (inv[0] unused)
    uint64_t inv_shoup[5];       // floor(inv[g] * 2^64 / M[g])
    uint64_t Mmod[5][5];         // Mmod[g][h] = M[h] mod M[g]            (h < g)
    uint64_t Mmod_shoup[5][5];   // Shoup companions
    uint64_t prefix_lo[6], prefix_hi[6];  // prefix[g] = M_0 ... M_{g-1} mod 2^128 (prefix[ngroups] = full product)
    uint32_t inv_shoup32[5];     // 32-bit Shoup companions, used when every digit modulus is one 30-bit prime
    uint32_t Mmod_shoup32[5][5];
};

// a * b mod m via Shoup (b < m < 2^63, a < 2^64): canonical
__device__ __forceinline__ uint64_t shoup_mulmod(uint64_t a, uint64_t b, uint64_t b_shoup, uint64_t m) {
    const uint64_t q = mulhi(a, b_shoup);
    const uint64_t r = a * b - q * m;
    return r >= m ? r - m : r;
}

struct u128d {
    uint64_t lo, hi;
};
__device__ __forceinline__ u128d mul_64x128(uint64_t a, uint64_t blo, uint64_t bhi) {  // a * b mod 2^128
    u128d r;
    r.lo = a * blo;
    r.hi = mulhi(a, blo) + a * bhi;
    return r;
}
__device__ __forceinline__ u128d add128(u128d a, u128d b) {
    u128d r;
    r.lo = a.lo + b.lo;
    r.hi = a.hi + b.hi + (r.lo < a.lo ? 1u : 0u);
    return r;
}
__device__ __forceinline__ u128d sub128(u128d a, u128d b) {
    u128d r;
    r.lo = a.lo - b.lo;
    r.hi = a.hi - b.hi - (a.lo < b.lo ? 1u : 0u);
    return r;
}

// a * b mod m for 30-bit m via Shoup (any 32-bit a): canonical
__device__ __forceinline__ uint32_t shoup_mulmod32(uint32_t a, uint32_t b, uint32_t b_shoup, uint32_t m) {
    const uint32_t r = a * b - __umulhi(a, b_shoup) * m;
    return umin<uint32_t>(r, r - m);
}

// NG digits; bit g of PAIRS set <=> digit g is a pair of primes.  Everything indexed by g / h is unrolled.
template <class W, class R, int NG, uint32_t PAIRS>
__global__ __launch_bounds__(256) void crt_kernel(W *__restrict__ value, CrtArgs A, size_t count) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    constexpr bool ALL32 = (PAIRS == 0u) && (sizeof(R) == 4);  // every digit modulus is a single 30-bit prime
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += stride) {
        uint64_t rg[NG];
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            const uint64_t va = reinterpret_cast<const R *>(A.res[A.ga[g]])[i];
            if (((PAIRS >> g) & 1u) != 0u) {  // compile-time after unrolling
                // v_b = (m_b - v_a) * P_a^-1 mod P_b ; group residue = v_a + v_b * P_a  (src/native64.rs:98-107)
                const uint64_t pa = A.prime[A.ga[g]], pb = A.prime[A.gb[g]];
                const uint64_t mb = reinterpret_cast<const R *>(A.res[A.gb[g]])[i];
                const uint32_t d = (uint32_t)(2 * pb + mb - va);  // < 3 * 2^30
                const uint64_t vb = shoup_mulmod32(d, (uint32_t)A.pair_inv[g], A.pair_inv_shoup[g], (uint32_t)pb);
                rg[g] = va + vb * pa;
            } else {
                rg[g] = va;
            }
        }
        // mixed-radix digits: v_g = (r_g - (v_0 + M_0 (v_1 + M_1 (...)))) * inv_g mod M_g
        uint64_t v[NG];
        v[0] = rg[0];
#pragma unroll
        for (int g = 1; g < NG; ++g) {
            if constexpr (ALL32) {
                const uint32_t m = (uint32_t)A.M[g];
                uint32_t acc = (uint32_t)v[g - 1];  // digits are < 2^30 < 2m: one conditional subtraction canonicalises
                acc = umin<uint32_t>(acc, acc - m);
#pragma unroll
                for (int h = g - 2; h >= 0; --h) {
                    uint32_t t = shoup_mulmod32(acc, (uint32_t)A.Mmod[g][h], A.Mmod_shoup32[g][h], m);
                    uint32_t vh = (uint32_t)v[h];
                    vh = umin<uint32_t>(vh, vh - m);
                    t += vh;
                    acc = umin<uint32_t>(t, t - m);
                }
                const uint32_t rr = (uint32_t)rg[g];
                const uint32_t d = rr - acc + m;  // in (0, 2m)
                v[g] = shoup_mulmod32(d, (uint32_t)A.inv[g], A.inv_shoup32[g], m);
            } else {
                // The digit moduli of every reference plan ascend (host.hip build_crt_args checks it), so each digit
                // v[h] < M[h] < M[g] and the group residue rg[g] < M[g] are canonical modulo M[g] as they are.  (A guarded
                // 64-bit `%` here expands to a long-division routine even though it never runs.)
                const uint64_t m = A.M[g];
                uint64_t acc = v[g - 1];
#pragma unroll
                for (int h = g - 2; h >= 0; --h) {
                    uint64_t t = shoup_mulmod(acc, A.Mmod[g][h], A.Mmod_shoup[g][h], m);
                    t += v[h];
                    acc = t >= m ? t - m : t;
                }
                const uint64_t rr = rg[g];
                const uint64_t d = rr >= acc ? rr - acc : rr + m - acc;
                v[g] = shoup_mulmod(d, A.inv[g], A.inv_shoup[g], m);
            }
        }
        const bool sign = v[NG - 1] > (A.M[NG - 1] / 2);  // centred lift decided by the TOP digit
        u128d pos = {v[0], 0};
#pragma unroll
        for (int g = 1; g < NG; ++g) pos = add128(pos, mul_64x128(v[g], A.prefix_lo[g], A.prefix_hi[g]));
        const u128d full = {A.prefix_lo[NG], A.prefix_hi[NG]};
        const u128d out = sign ? sub128(pos, full) : pos;
        if constexpr (sizeof(W) == 16) {
            reinterpret_cast<uint64_t *>(value)[2 * i] = out.lo;
            reinterpret_cast<uint64_t *>(value)[2 * i + 1] = out.hi;
        } else {
            value[i] = (W)out.lo;
        }
    }
}

// ---------------------------------------------------------------------------------------------
// product::Plan (src/product.rs): residue split of u64 coefficients and Garner recombination.
// Batched NTT-domain layout: plane-major.  u32 plane k (k < n32) holds `count` = batch * n residues at
// res32 + k * count; u64 plane k at res64 + k * count, where res64 = res32 + n32 * count u32 words.
// With batch == 1 this is exactly the reference's layout (src/product.rs:261-270).
// Both kernels are elementwise and HBM-bound: 8 B in, 4 B * n32 + 8 B * n64 out per coefficient.
// ---------------------------------------------------------------------------------------------
constexpr int PRODUCT_MAX_PRIMES = 8;  // distinct primes = 1 mod 64 with a product < 2^64: at most 7

struct ProductArgs {
    int n32, n64;
    uint64_t modulus, bound;
    uint64_t prime[PRODUCT_MAX_PRIMES];    // ascending: the n32 primes below 2^32 first (src/product.rs:183-184)
    uint64_t barrett[PRODUCT_MAX_PRIMES];  // floor(2^64 / prime)
    uint64_t inv[28], inv_shoup[28];       // pair (j, i < j) at j(j-1)/2 + i: prime[i]^-1 mod prime[j] (src/product.rs:207-229)
};

// x mod p for any odd p < 2^64, no division: q = floor(x * floor(2^64/p) / 2^64) >= floor(x/p) - 1
__device__ __forceinline__ uint64_t barrett_rem(uint64_t x, uint64_t p, uint64_t m) {
    uint64_t r = x - mulhi(x, m) * p;
    r = r >= p ? r - p : r;
    return r >= p ? r - p : r;
}

// MODE 0: FwdMode::Generic, `%` per prime (src/product.rs:323-355)
// MODE 1: FwdMode::Bounded fast path of the u32x2 plan (src/product.rs:303-322), same select as the reference
// MODE 2: single-prime plans: u64x1 copies, u32x1 truncates -- no reduction (src/product.rs:282-293)
template <int MODE>
__global__ __launch_bounds__(256) void product_split_kernel(uint32_t *__restrict__ res32, uint64_t *__restrict__ res64,
                                                            const uint64_t *__restrict__ standard, ProductArgs A,
                                                            size_t count) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    const size_t pairs = count / 2;  // n is even (src/product.rs:158)
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < pairs; i += stride) {
        const ulonglong2 s = reinterpret_cast<const ulonglong2 *>(standard)[i];
        if constexpr (MODE == 1) {
            const uint32_t p0 = (uint32_t)A.prime[0], p1 = (uint32_t)A.prime[1], pu = (uint32_t)A.modulus;
            const uint64_t half = A.modulus / 2;
            const uint32_t sx = (uint32_t)s.x, sy = (uint32_t)s.y;
            const bool posx = s.x < half, posy = s.y < half;
            const uint32_t cx = pu - sx, cy = pu - sy;
            reinterpret_cast<uint2 *>(res32)[i] = make_uint2(posx ? sx : p0 - cx, posy ? sy : p0 - cy);
            reinterpret_cast<uint2 *>(res32 + count)[i] = make_uint2(posx ? sx : p1 - cx, posy ? sy : p1 - cy);
        } else if constexpr (MODE == 2) {
            if (A.n32 == 1) reinterpret_cast<uint2 *>(res32)[i] = make_uint2((uint32_t)s.x, (uint32_t)s.y);
            else reinterpret_cast<ulonglong2 *>(res64)[i] = s;
        } else {
            for (int k = 0; k < A.n32; ++k)
                reinterpret_cast<uint2 *>(res32 + (size_t)k * count)[i] =
                    make_uint2((uint32_t)barrett_rem(s.x, A.prime[k], A.barrett[k]),
                               (uint32_t)barrett_rem(s.y, A.prime[k], A.barrett[k]));
            for (int k = 0; k < A.n64; ++k) {
                const uint64_t p = A.prime[A.n32 + k], m = A.barrett[A.n32 + k];
                ulonglong2 r;
                r.x = barrett_rem(s.x, p, m);
                r.y = barrett_rem(s.y, p, m);
                reinterpret_cast<ulonglong2 *>(res64 + (size_t)k * count)[i] = r;
            }
        }
    }
}

__device__ __forceinline__ uint64_t add_mod_u64(uint64_t m, uint64_t a, uint64_t b) {  // src/product.rs:85-92
    const uint64_t sum = a + b;
    return (sum >= m || sum < a) ? sum - m : sum;
}
__device__ __forceinline__ uint32_t add_mod_u32(uint32_t m, uint32_t a, uint32_t b) {  // src/product.rs:107-114
    const uint32_t sum = a + b;
    return (sum >= m || sum < a) ? sum - m : sum;
}

// Garner (Knuth 4.3.2) over K residues of one coefficient, then Horner: src/product.rs:791-879 (and its
// u64x1 / u32x1 / u32x2 special cases :386-789, which produce the same digits).
// ACC 0: InvMode::Replace; 1: Accumulate with add_mod_u64(modulus, ..); 2: Accumulate of the u32x1 plan, which
// the reference performs in u32 on the truncated `standard` (src/product.rs:408-413).
template <int K> __device__ __forceinline__ uint64_t garner(const uint64_t (&u)[K], const ProductArgs &A) {
    uint64_t v[K];
#pragma unroll
    for (int j = 0; j < K; ++j) {
        uint64_t x = u[j];
        const uint64_t pj = A.prime[j];
#pragma unroll
        for (int i = 0; i < j; ++i) {
            const uint64_t d = x >= v[i] ? x - v[i] : x - v[i] + pj;  // sub_mod: v[i] < prime[i] < prime[j]
            x = shoup_mulmod(d, A.inv[j * (j - 1) / 2 + i], A.inv_shoup[j * (j - 1) / 2 + i], pj);
        }
        v[j] = x;
    }
    uint64_t acc = 0;
#pragma unroll
    for (int j = K - 1; j >= 0; --j) acc = acc * A.prime[j] + v[j];
    return acc;
}

template <int K, int ACC>
__global__ __launch_bounds__(256) void product_crt_kernel(uint64_t *__restrict__ standard, const uint32_t *__restrict__ res32,
                                                          const uint64_t *__restrict__ res64, ProductArgs A, size_t count) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    const size_t pairs = count / 2;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < pairs; i += stride) {
        uint64_t ux[K], uy[K];
#pragma unroll
        for (int j = 0; j < K; ++j) {
            if (j < A.n32) {
                const uint2 r = reinterpret_cast<const uint2 *>(res32 + (size_t)j * count)[i];
                ux[j] = r.x;
                uy[j] = r.y;
            } else {
                const ulonglong2 r = reinterpret_cast<const ulonglong2 *>(res64 + (size_t)(j - A.n32) * count)[i];
                ux[j] = r.x;
                uy[j] = r.y;
            }
        }
        ulonglong2 out;
        out.x = garner<K>(ux, A);
        out.y = garner<K>(uy, A);
        if constexpr (ACC == 1) {
            const ulonglong2 s = reinterpret_cast<const ulonglong2 *>(standard)[i];
            out.x = add_mod_u64(A.modulus, s.x, out.x);
            out.y = add_mod_u64(A.modulus, s.y, out.y);
        } else if constexpr (ACC == 2) {
            const ulonglong2 s = reinterpret_cast<const ulonglong2 *>(standard)[i];
            out.x = add_mod_u32((uint32_t)A.modulus, (uint32_t)s.x, (uint32_t)out.x);
            out.y = add_mod_u32((uint32_t)A.modulus, (uint32_t)s.y, (uint32_t)out.y);
        }
        reinterpret_cast<ulonglong2 *>(standard)[i] = out;
    }
}

}  // namespace cntt
