// libcntt_hip.so host side: plans, device tables, launch logic and the C ABI of include/cntt.h.
// There is no CPU compute path in this library: every transform runs in the HIP kernels, and every
// entry point that needs a GPU fails with CNTT_EDEVICE when none is present.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <memory>
#include <mutex>
#include <new>
#include <string>
#include <vector>

#include "../../include/cntt.h"
#include "aux_kernels.hpp"
#include "host_math.hpp"
#include "native_fused.hpp"
#include "product_fused.hpp"
#include "ntt_launch.hpp"

using namespace cntt;
using host::u128;

// ---------------------------------------------------------------------------------------------
// errors
// ---------------------------------------------------------------------------------------------
static thread_local std::string g_err;

static int fail(int code, const char *fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_err = buf;
    return code;
}
#define HIP_TRY(expr)                                                                              \
    do {                                                                                           \
        hipError_t e_ = (expr);                                                                    \
        if (e_ != hipSuccess) return fail(CNTT_EDEVICE, "%s: %s", #expr, hipGetErrorString(e_));    \
    } while (0)

extern "C" const char *cntt_last_error(void) { return g_err.c_str(); }
#include "build_hash.inc"  // CNTT_CSRC_HASH: sha256 of the csrc/ sources this library was built from (Makefile)
extern "C" const char *cntt_version(void) { return "cntt-hip 0.3 (gfx950) csrc:" CNTT_CSRC_HASH; }
extern "C" int cntt_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

// ---------------------------------------------------------------------------------------------
// testing only: the kernel-selection switchboard (ntt_launch.hpp DebugSwitch).  Plain atomics: set from a test or an A/B tool through
// cntt_debug_set(), read where a path is chosen (the two class switches at plan creation, the rest at the call).  Nothing here, or
// anywhere else in the library, reads the process environment.
// ---------------------------------------------------------------------------------------------
#include <atomic>
namespace {
struct SwitchDef { const char *name; int dflt; };
constexpr SwitchDef kSwitches[DBG_COUNT] = {
    {"fp", 1}, {"pm64", 1}, {"blk", 1}, {"mul32_blk", 1}, {"ext32_blk", 1}, {"ext_one", 1}, {"ext_split", -1}, {"native_acc", 1},
    {"product_fused", -1}, {"plan52_via32", 1}};
std::atomic<int> g_switch[DBG_COUNT] = {{1}, {1}, {1}, {1}, {1}, {1}, {-1}, {1}, {-1}, {1}};
int switch_index(const char *key) {
    if (!key) return -1;
    for (int i = 0; i < (int)DBG_COUNT; ++i)
        if (std::strcmp(key, kSwitches[i].name) == 0) return i;
    return -1;
}
}  // namespace
int cntt::debug_switch(DebugSwitch key) { return g_switch[key].load(std::memory_order_relaxed); }
extern "C" int cntt_debug_set(const char *key, int value) {
    if (key && std::strcmp(key, "reset") == 0) {
        for (int i = 0; i < (int)DBG_COUNT; ++i) g_switch[i].store(kSwitches[i].dflt);
        return CNTT_OK;
    }
    const int i = switch_index(key);
    if (i < 0) return fail(CNTT_EINVAL, "cntt_debug_set: unknown switch '%s'", key ? key : "(null)");
    if (value < -1 || value > 1) return fail(CNTT_EINVAL, "cntt_debug_set: %s takes -1 (library default), 0 or 1", key);
    g_switch[i].store(value < 0 ? kSwitches[i].dflt : value);
    return CNTT_OK;
}
extern "C" int cntt_debug_get(const char *key, int *value) {
    const int i = switch_index(key);
    if (i < 0 || !value) return fail(CNTT_EINVAL, "cntt_debug_get: unknown switch '%s'", key ? key : "(null)");
    *value = g_switch[i].load();
    return CNTT_OK;
}

// ---------------------------------------------------------------------------------------------
// batch partition over the devices of a node: contiguous shards, remainders to the low ranks -- the arithmetic of
// concrete-ntt_amd/shard.py shard_bounds(), for C / Rust callers that drive several devices themselves (examples/multi_device.cpp)
// ---------------------------------------------------------------------------------------------
extern "C" int cntt_shard_bounds(size_t batch, int world, int rank, size_t *begin, size_t *end) {
    if (world < 1 || rank < 0 || rank >= world || !begin || !end) return fail(CNTT_EINVAL, "cntt_shard_bounds: need 0 <= rank < world and non-NULL outputs");
    const size_t base = batch / (size_t)world, rem = batch % (size_t)world, r = (size_t)rank;
    *begin = r * base + (r < rem ? r : rem);
    *end = *begin + base + (r < rem ? 1 : 0);
    return CNTT_OK;
}

// Grid of the element-wise kernels (grid-stride loops: any grid is correct): one 256-thread block per 256 work items, NOT capped at
// a few blocks per CU.  Measured (tools/pw_probe.py, 65536 x 1024 u64): mul_assign_normalize 0.374 -> 0.271 ms (4.3 -> 5.9 TB/s),
// normalize 0.232 -> 0.178 ms (4.6 -> 6.0 TB/s) against the 2048-block persistent form; split / CRT / Garner kernels -4 ... -7 %.
// The cap: a dispatch holds at most 2^32 - 1 work-items per dimension (grid_size_x of the AQL packet is a uint32), i.e. fewer than
// 2^24 blocks of 256 threads; beyond it (16 GiB of u32 and up -- plausible on 288 GB) the kernels' 64-bit grid-stride loops
// take over.
extern "C" unsigned cntt_ew_grid(size_t work_items) {
    const size_t blocks = (work_items + 255) / 256;
    return (unsigned)std::max<size_t>(1, std::min<size_t>(blocks, ((size_t)1 << 24) - 1));
}
static inline unsigned ew_grid(size_t work_items) { return cntt_ew_grid(work_items); }

// ---------------------------------------------------------------------------------------------
// prime plans
// ---------------------------------------------------------------------------------------------
template <class T> struct DeviceTables {
    TwPair<T> *fwd = nullptr, *inv = nullptr;
    // plans whose LDS-resident transforms run in a 64-bit-only class: the same twiddles in that class's form --
    // CLS_FP / CLS_FP51: (c, c / p) doubles, c centred in (-p/2, p/2];  CLS_PM64: plain residues;  CLS_FPW: c as one double
    TwPair<T> *fwd_fp = nullptr, *inv_fp = nullptr;
};

template <class T> struct DeviceCache {
    std::mutex mu;
    std::map<int, DeviceTables<T>> per_device;
    ~DeviceCache() {
        for (auto &kv : per_device) {
            int cur = 0;
            if (hipGetDevice(&cur) != hipSuccess) continue;
            (void)hipSetDevice(kv.first);
            (void)hipFree(kv.second.fwd);
            (void)hipFree(kv.second.inv);
            (void)hipFree(kv.second.fwd_fp);
            (void)hipFree(kv.second.inv_fp);
            (void)hipSetDevice(cur);
        }
    }
};

template <class T> struct PrimePlan {
    static constexpr int B = sizeof(T) * 8;
    size_t n = 0;
    int logn = 0;
    T p = 0;
    uint64_t root = 0;
    bool has_shoup = false;
    // reference-layout tables (src/prime64.rs:221-236)
    std::vector<T> twid, twid_shoup, inv_twid, inv_twid_shoup;
    T p_barrett = 0, n_inv = 0, n_inv_shoup = 0;
    uint32_t big_q = 0;
    ModParams<T> mp{};
    std::shared_ptr<DeviceCache<T>> cache;
};

struct cntt_plan64 : PrimePlan<uint64_t> {};
struct cntt_plan32 : PrimePlan<uint32_t> {};

template <class T> static T shoup_of(T x, T p) { return (T)((((u128)x) << (sizeof(T) * 8)) / p); }

// Plan::try_new  (src/prime64.rs:704-771, src/prime32.rs:630-686)
template <class T, class PlanT> static int plan_new(size_t n, T p, PlanT **out) {
    constexpr int B = sizeof(T) * 8;
    if (!out) return fail(CNTT_EINVAL, "out is NULL");
    *out = nullptr;
    if (p <= 1) return fail(CNTT_EINVAL, "modulus <= 1: the reference panics in Div%d::new (src/fastdiv.rs)", B);
    const size_t min_n = (B == 64) ? 16 : 32;
    if (n < min_n || (n & (n - 1)) != 0) return fail(CNTT_NONE, "polynomial_size must be a power of two >= %zu", min_n);
    if (n > ((size_t)1 << 30)) return fail(CNTT_NONE, "polynomial_size too large");
    if (!host::is_prime((uint64_t)p)) return fail(CNTT_NONE, "modulus is not prime");
    uint64_t w = 0;
    if (!host::primitive_root((uint64_t)p, 2 * (uint64_t)n, &w))
        return fail(CNTT_NONE, "no primitive 2n-th root of unity modulo the modulus");

    PlanT *pl = new (std::nothrow) PlanT();
    if (!pl) return fail(CNTT_ENOMEM, "out of memory");
    pl->n = n;
    pl->p = p;
    pl->root = w;
    while (((size_t)1 << pl->logn) < n) ++pl->logn;
    pl->has_shoup = (uint64_t)p < ((uint64_t)1 << (B - 1));
    pl->twid.assign(n, 0);
    pl->inv_twid.assign(n, 0);
    if (pl->has_shoup) {
        pl->twid_shoup.assign(n, 0);
        pl->inv_twid_shoup.assign(n, 0);
    }
    // twid[bitrev(k)] = w^k ; inv_twid[bitrev((n-k) mod n)] = (k == 0 ? 1 : p - w^k)
    uint64_t wk = 1;
    for (size_t k = 0; k < n; ++k) {
        const size_t fi = host::bit_reverse((uint32_t)pl->logn, (uint32_t)k);
        const size_t ii = host::bit_reverse((uint32_t)pl->logn, (uint32_t)((n - k) % n));
        const T x = (k == 0) ? (T)wk : (T)(p - (T)wk);
        pl->twid[fi] = (T)wk;
        pl->inv_twid[ii] = x;
        if (pl->has_shoup) {
            pl->twid_shoup[fi] = shoup_of<T>((T)wk, p);
            pl->inv_twid_shoup[ii] = shoup_of<T>(x, p);
        }
        wk = host::mulmod(wk, w, (uint64_t)p);
    }
    pl->n_inv = (T)host::powmod((uint64_t)n % p, (uint64_t)p - 2, (uint64_t)p);
    pl->n_inv_shoup = shoup_of<T>(pl->n_inv, p);
    uint32_t ilog = 0;
    while (ilog + 1 < (uint32_t)B && (((uint64_t)p) >> (ilog + 1)) != 0) ++ilog;
    pl->big_q = ilog + 1;
    {
        const uint32_t big_l = pl->big_q + (B - 1);
        pl->p_barrett = (T)((((u128)1) << big_l) / p);  // unused garbage when p >= 2^(B-1), as in the reference
        if (big_l >= 128) pl->p_barrett = 0;
    }
    // device-side parameters
    ModParams<T> &mp = pl->mp;
    mp.p = p;
    mp.neg_p = (T)0 - p;
    mp.two_p = (T)(2 * p);
    mp.neg_two_p = (T)0 - (T)(2 * p);
    mp.big_q = pl->big_q;
    mp.p_barrett = pl->p_barrett;
    const uint64_t p64 = (uint64_t)p;
    if (p64 < ((uint64_t)1 << (B - 2)))
        mp.cls = CLS_LAZY;
    else if (p64 < ((uint64_t)1 << (B - 1)))
        mp.cls = CLS_STRICT;
    else
        mp.cls = CLS_GENERIC;
    mp.pinv_neg = (T)host::neg_inv_pow2(p64);
    const u128 R = ((u128)1) << B;
    const uint64_t r1 = (uint64_t)(R % p64);
    const uint64_t r2 = host::mulmod(r1, r1, p64);
    mp.r2 = (T)r2;
    const uint64_t w_last = host::mulmod((uint64_t)pl->inv_twid[1], (uint64_t)pl->n_inv, p64);  // inv_twid[1] / N
    if (mp.cls == CLS_GENERIC) {
        mp.n_inv = (T)host::mulmod((uint64_t)pl->n_inv, r2, p64);  // N^-1 * R^2 (see mul_normalize)
        mp.n_inv_shoup = 0;
        mp.last_w = (T)host::mulmod(w_last, r2, p64);
        mp.last_w_shoup = 0;
    } else {
        mp.n_inv = pl->n_inv;
        mp.n_inv_shoup = pl->n_inv_shoup;
        mp.last_w = (T)w_last;
        mp.last_w_shoup = shoup_of<T>((T)w_last, p);
        // the same two constants times 2^B: the fused product kernels of the lazy class multiply pointwise with a Montgomery product
        // (ntt_arith.hpp mul_fused), whose 2^-B is undone where 1/N is applied
        mp.mont_n_inv = (T)host::mulmod((uint64_t)pl->n_inv, r1, p64);
        mp.mont_n_inv_shoup = shoup_of<T>(mp.mont_n_inv, p);
        mp.mont_last_w = (T)host::mulmod(w_last, r1, p64);
        mp.mont_last_w_shoup = shoup_of<T>(mp.mont_last_w, p);
        mp.mont_r = (T)r1;   // the fused mul_accumulate chains multiply their accumulators by 2^B once (ntt_arith.hpp chain_pre_inverse)
        mp.mont_r_shoup = shoup_of<T>(mp.mont_r, p);
    }
    // CLS_FP / CLS_FP51: 64-bit words, p < 2^50 / 2^51 (the classes of src/prime64/less_than_50bit.rs and
    // less_than_51bit.rs).  cntt_debug_set("fp", 0) keeps such plans on the integer butterflies (A/B measurements, and tests
    // that compare the two paths).
    mp.fp = 0;
    if constexpr (B == 64) {
        if (p64 < ((uint64_t)1 << 51) && debug_switch(DBG_FP) != 0) {
            mp.fp = p64 < ((uint64_t)1 << 50) ? (uint32_t)CLS_FP : (uint32_t)CLS_FP51;
            const double pd = (double)p64;
            mp.fp_p = host::double_bits(pd);
            mp.fp_pinv = host::double_bits(1.0 / pd);
            const double ni = host::centred(pl->n_inv, p64), wl = host::centred(w_last, p64);
            mp.fp_n_inv = host::double_bits(ni);
            mp.fp_n_inv_q = host::double_bits(ni / pd);
            mp.fp_last_w = host::double_bits(wl);
            mp.fp_last_w_q = host::double_bits(wl / pd);
        }
    }
    // CLS_FPW: 32-bit words, p >= 2^31 (no lazy headroom in 32 bits: the Montgomery class otherwise): the LDS-resident
    // transforms run on doubles (ntt_arith.hpp).  cntt_debug_set("fp", 0) keeps the Montgomery class here too.
    if constexpr (B == 32) {
        if (mp.cls == CLS_GENERIC && debug_switch(DBG_FP) != 0) {
            mp.fp = (uint32_t)CLS_FPW;
            const double pd = (double)p64;
            mp.fp_p = host::double_bits(pd);
            mp.fp_pinv = host::double_bits(1.0 / pd);
            const double ni = host::centred(pl->n_inv, p64), wl = host::centred(w_last, p64);
            mp.fp_n_inv = host::double_bits(ni);
            mp.fp_n_inv_q = host::double_bits(ni / pd);
            mp.fp_last_w = host::double_bits(wl);
            mp.fp_last_w_q = host::double_bits(wl / pd);
        }
    }
    // CLS_PM64: p = 2^64 - c with c < 2^32 (Solinas, and the largest primes below 2^64).  cntt_debug_set("pm64", 0) keeps the
    // Montgomery class (A/B measurements and tests).
    mp.pm_c = 0;
    if constexpr (B == 64) {
        const uint64_t c = (uint64_t)0 - p64;
        if (p64 >= ((uint64_t)1 << 63) && c < ((uint64_t)1 << 32) && debug_switch(DBG_PM64) != 0) {
            mp.pm_c = (uint32_t)c;
            mp.pm_n_inv = pl->n_inv;
            mp.pm_last_w = (T)w_last;
        }
    }
    pl->cache = std::make_shared<DeviceCache<T>>();
    *out = pl;
    return CNTT_OK;
}

static bool alt_tables(int cls) { return is_fp_class(cls) || cls == CLS_PM64 || cls == CLS_FPW; }

// transform class of the LDS-resident kernels for this plan (the global-stage path of larger sizes is integer-only)
template <class T> static int transform_class(const PrimePlan<T> *pl) {
    if (pl->logn > MaxLdsLogN<T>::value) return (int)pl->mp.cls;
    if (pl->mp.fp == CLS_FPW && pl->logn > MAX_FPW_LOGN) return (int)pl->mp.cls;
    if (pl->mp.fp) return (int)pl->mp.fp;
    if (pl->mp.pm_c) return (int)CLS_PM64;
    return (int)pl->mp.cls;
}

// per-device table replica, created on first use under the cache mutex
template <class T> static int device_tables(const PrimePlan<T> *pl, DeviceTables<T> *out) {
    int dev = 0;
    HIP_TRY(hipGetDevice(&dev));
    std::lock_guard<std::mutex> lk(pl->cache->mu);
    auto it = pl->cache->per_device.find(dev);
    if (it != pl->cache->per_device.end()) {
        *out = it->second;
        return CNTT_OK;
    }
    const size_t n = pl->n;
    std::vector<TwPair<T>> f(n), i(n);
    const uint64_t p64 = (uint64_t)pl->p;
    const uint64_t r1 = (uint64_t)((((u128)1) << (sizeof(T) * 8)) % p64);
    for (size_t k = 0; k < n; ++k) {
        if (pl->mp.cls == CLS_GENERIC) {  // Montgomery form
            f[k].w = (T)host::mulmod((uint64_t)pl->twid[k], r1, p64);
            f[k].ws = 0;
            i[k].w = (T)host::mulmod((uint64_t)pl->inv_twid[k], r1, p64);
            i[k].ws = 0;
        } else {
            f[k].w = pl->twid[k];
            f[k].ws = pl->twid_shoup[k];
            i[k].w = pl->inv_twid[k];
            i[k].ws = pl->inv_twid_shoup[k];
        }
    }
    DeviceTables<T> t;
    // every early return below (allocation or upload failure) releases what this call allocated so far
    struct Release {
        DeviceTables<T> *t;
        ~Release() {
            if (!t) return;
            (void)hipFree(t->fwd);
            (void)hipFree(t->inv);
            (void)hipFree(t->fwd_fp);
            (void)hipFree(t->inv_fp);
        }
    } release{&t};
    if (hipMalloc((void **)&t.fwd, n * sizeof(TwPair<T>)) != hipSuccess || hipMalloc((void **)&t.inv, n * sizeof(TwPair<T>)) != hipSuccess)
        return fail(CNTT_ENOMEM, "hipMalloc of the twiddle tables failed");
    HIP_TRY(hipMemcpy(t.fwd, f.data(), n * sizeof(TwPair<T>), hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(t.inv, i.data(), n * sizeof(TwPair<T>), hipMemcpyHostToDevice));
    if (pl->mp.fp || pl->mp.pm_c) {
        const double pd = (double)p64;
        for (size_t k = 0; k < n; ++k) {
            if constexpr (sizeof(T) == 8) {
                if (pl->mp.fp) {
                    const double cf = host::centred((uint64_t)pl->twid[k], p64), ci = host::centred((uint64_t)pl->inv_twid[k], p64);
                    f[k].w = host::double_bits(cf);
                    f[k].ws = host::double_bits(cf / pd);
                    i[k].w = host::double_bits(ci);
                    i[k].ws = host::double_bits(ci / pd);
                } else {
                    f[k].w = pl->twid[k];
                    f[k].ws = 0;
                    i[k].w = pl->inv_twid[k];
                    i[k].ws = 0;
                }
            } else {  // CLS_FPW: the 8-byte entry is the centred twiddle as a double (w = low, ws = high word)
                (void)pd;
                const uint64_t bf = host::double_bits(host::centred((uint64_t)pl->twid[k], p64));
                const uint64_t bi = host::double_bits(host::centred((uint64_t)pl->inv_twid[k], p64));
                f[k].w = (T)bf;
                f[k].ws = (T)(bf >> 32);
                i[k].w = (T)bi;
                i[k].ws = (T)(bi >> 32);
            }
        }
        if (hipMalloc((void **)&t.fwd_fp, n * sizeof(TwPair<T>)) != hipSuccess ||
            hipMalloc((void **)&t.inv_fp, n * sizeof(TwPair<T>)) != hipSuccess)
            return fail(CNTT_ENOMEM, "hipMalloc of the twiddle tables failed");
        HIP_TRY(hipMemcpy(t.fwd_fp, f.data(), n * sizeof(TwPair<T>), hipMemcpyHostToDevice));
        HIP_TRY(hipMemcpy(t.inv_fp, i.data(), n * sizeof(TwPair<T>), hipMemcpyHostToDevice));
    }
    release.t = nullptr;  // the cache owns the tables from here
    pl->cache->per_device[dev] = t;
    *out = t;
    return CNTT_OK;
}

template <class T, bool INV, int CLS>
static void launch_global_stage(T *data, const TwPair<T> *tw, const ModParams<T> &P, uint32_t logn, uint32_t s,
                                size_t nb, bool finish, hipStream_t st) {
    hipLaunchKernelGGL((global_stage_kernel<T, INV, CLS>), dim3(ew_grid(nb)), dim3(256), 0, st, data, tw, P, logn, s, nb,
                       finish);
}
template <class T, bool INV>
static void global_stage(T *data, const TwPair<T> *tw, const ModParams<T> &P, uint32_t logn, uint32_t s, size_t nb,
                         bool finish, hipStream_t st) {
    switch (P.cls) {
    case CLS_LAZY: launch_global_stage<T, INV, CLS_LAZY>(data, tw, P, logn, s, nb, finish, st); break;
    case CLS_STRICT: launch_global_stage<T, INV, CLS_STRICT>(data, tw, P, logn, s, nb, finish, st); break;
    default: launch_global_stage<T, INV, CLS_GENERIC>(data, tw, P, logn, s, nb, finish, st); break;
    }
}

// batched transform on device memory
template <class T> static int ntt_device(const PrimePlan<T> *pl, T *d, size_t batch, bool inv, hipStream_t st) {
    if (batch == 0) return CNTT_OK;
    DeviceTables<T> t;
    if (int rc = device_tables(pl, &t)) return rc;
    const int maxl = MaxLdsLogN<T>::value;
    const int depth = pl->logn > maxl ? pl->logn - maxl : 0;
    const int sub_logn = pl->logn - depth;
    if ((batch << depth) >= ((size_t)1 << 32)) return fail(CNTT_EINVAL, "batch too large for one launch");
    const uint32_t nsub = (uint32_t)(batch << depth);
    const size_t nbfly = batch * (pl->n / 2);
    const int tcls = transform_class(pl);
    // per-launch copy of the plan's parameters: a batch beyond the 256 MiB Infinity Cache streams (ModParams::stream, ntt_kernel.hpp)
    ModParams<T> mp = pl->mp;
    mp.stream = batch * pl->n * sizeof(T) > ((size_t)384 << 20) ? 1u : 0u;
    hipError_t e;
    if (depth == 1 && batch < ((size_t)1 << 32)) {
        // one size past the LDS-resident ones: a single-pass kernel exists for 64-bit words (Ntt32k), in the plan's
        // 64-bit-only class where it has one (those tables exist at every size)
        int c1 = tcls;
        if constexpr (sizeof(T) == 8) c1 = pl->mp.fp ? (int)pl->mp.fp : pl->mp.pm_c ? (int)CLS_PM64 : tcls;
        e = inv ? launch_ntt<T, true>(pl->logn, c1, d, alt_tables(c1) ? t.inv_fp : t.inv, mp, (uint32_t)batch, 0u, st)
                : launch_ntt<T, false>(pl->logn, c1, d, alt_tables(c1) ? t.fwd_fp : t.fwd, mp, (uint32_t)batch, 0u, st);
        if (e == hipSuccess) return CNTT_OK;
        (void)hipGetLastError();
        if (e != hipErrorNotSupported && e != hipErrorInvalidValue)
            return fail(CNTT_EDEVICE, "NTT kernel launch failed: %s", hipGetErrorString(e));
    }
    if (!inv) {
        for (int s = 0; s < depth; ++s) global_stage<T, false>(d, t.fwd, pl->mp, (uint32_t)pl->logn, (uint32_t)s, nbfly, false, st);
        e = launch_ntt<T, false>(sub_logn, tcls, d, alt_tables(tcls) ? t.fwd_fp : t.fwd, mp, nsub, (uint32_t)depth, st);
    } else {
        e = launch_ntt<T, true>(sub_logn, tcls, d, alt_tables(tcls) ? t.inv_fp : t.inv, mp, nsub, (uint32_t)depth, st);
        for (int s = depth - 1; s >= 0 && e == hipSuccess; --s)
            global_stage<T, true>(d, t.inv, pl->mp, (uint32_t)pl->logn, (uint32_t)s, nbfly, s == 0, st);
    }
    if (e != hipSuccess) return fail(CNTT_EDEVICE, "NTT kernel launch failed: %s", hipGetErrorString(e));
    HIP_TRY(hipGetLastError());
    return CNTT_OK;
}

template <class T, int OP>
static int pointwise_device(const PrimePlan<T> *pl, T *a, const T *b, const T *c, size_t count, hipStream_t st) {
    if (count == 0) return CNTT_OK;
    const size_t nv = count / (16 / sizeof(T)) + 1;
    // working set of the call against the 256 MiB Infinity Cache: larger ones stream (non-temporal policy, aux_kernels.hpp)
    constexpr size_t NARR = OP == PW_NORMALIZE ? 1 : OP == PW_MUL_ACCUMULATE ? 3 : 2;
    if (NARR * count * sizeof(T) > ((size_t)384 << 20))
        hipLaunchKernelGGL((pointwise_kernel<T, OP, true>), dim3(ew_grid(nv)), dim3(256), 0, st, a, b, c, pl->mp, count);
    else
        hipLaunchKernelGGL((pointwise_kernel<T, OP, false>), dim3(ew_grid(nv)), dim3(256), 0, st, a, b, c, pl->mp, count);
    HIP_TRY(hipGetLastError());
    return CNTT_OK;
}

// RAII device scratch for host-memory calls
struct DevBuf {
    void *p = nullptr;
    ~DevBuf() {
        if (p) (void)hipFree(p);
    }
    int alloc(size_t bytes) {
        if (hipMalloc(&p, bytes ? bytes : 16) != hipSuccess) {
            p = nullptr;
            return fail(CNTT_EDEVICE, "hipMalloc(%zu) failed (no GPU, or out of device memory)", bytes);
        }
        return CNTT_OK;
    }
};

// fused lhs <- inv(mul_assign_normalize(fwd(lhs), rhs_ntt)); three launches when no fused kernel exists
template <class T> static int mul_ntt_device(const PrimePlan<T> *pl, T *lhs, const T *rhs, size_t batch, hipStream_t st) {
    if (batch == 0) return CNTT_OK;
    if (batch >= ((size_t)1 << 32)) return fail(CNTT_EINVAL, "batch too large for one launch");
    DeviceTables<T> t;
    if (int rc = device_tables(pl, &t)) return rc;
    int tcls = transform_class(pl);
    // one size past the LDS-resident ones (64-bit words): the single-pass kernels run in the plan's 64-bit-only class where
    // it has one, like ntt_device
    if (sizeof(T) == 8 && pl->logn == MaxLdsLogN<T>::value + 1) tcls = pl->mp.fp ? (int)pl->mp.fp : pl->mp.pm_c ? (int)CLS_PM64 : tcls;
    const hipError_t e = launch_mul_ntt<T>(pl->logn, tcls, lhs, rhs, alt_tables(tcls) ? t.fwd_fp : t.fwd,
                                           alt_tables(tcls) ? t.inv_fp : t.inv, pl->mp, (uint32_t)batch, st);
    if (e == hipSuccess) return CNTT_OK;
    if (e != hipErrorNotSupported) return fail(CNTT_EDEVICE, "fused product launch failed: %s", hipGetErrorString(e));
    (void)hipGetLastError();
    if (int rc = ntt_device<T>(pl, lhs, batch, false, st)) return rc;
    if (int rc = pointwise_device<T, PW_MUL_NORMALIZE>(pl, lhs, rhs, nullptr, batch * pl->n, st)) return rc;
    return ntt_device<T>(pl, lhs, batch, true, st);
}

// Three / four outputs of the fused mul_accumulate chain where only the one- / two-output kernels exist (64-bit words at n = 16384,
// 32-bit words at n = 32768): two fused launches of <= 2 outputs against the composed path, measured per class with J = 6
// (profiles/r04_chain_split.jsonl, ms per 1024 elements, split / composed):
//   u64 n = 16384   p < 2^50 (doubles)  0.99 / 1.48, 1.12 / 1.61     2^64 - c  1.77 / 2.00, 1.98 / 2.31     -> split
//                   62-bit  1.74 / 1.61, 1.98 / 1.88     63-bit  1.87 / 1.70, 2.16 / 1.99                     -> composed
//   u32 n = 32768   30-bit  1.45 / 1.61, 1.73 / 1.94                                                          -> split
//                   31-bit  1.70 / 1.75, 2.01 / 1.97     p >= 2^31 (doubles)  2.14 / 2.07, 2.58 / 2.48        -> composed
// cntt_debug_set("ext_split", 0) never splits, 1 always does (A/B runs); the decision of the call in flight sits in a thread-local because
// launch_ext_ntt() (ntt_launch.hpp) only asks ext_split_enabled().
static thread_local bool g_ext_split_wins = true;
static bool ext_split_wins(size_t word, int logn, int cls) {
    if (word == 8 && logn == 14) return cls == CLS_FP || cls == CLS_FP51 || cls == CLS_PM64;
    if (word == 4 && logn == 15) return cls == CLS_LAZY;
    return true;
}
bool cntt::ext_split_enabled() {
    const int mode = debug_switch(DBG_EXT_SPLIT);
    return mode < 0 ? g_ext_split_wins : mode == 1;
}
// mul_accumulate chain: out[b][o] (+)= inv(sum_j fwd(terms[b][j]) . key_ntt[j][o])  (device pointers).
// Fused kernel when the transform lives in one wavefront group and nout <= 4; otherwise composed from the batched
// kernels through a stream-ordered scratch allocation.
template <class T>
static int external_product_device(const PrimePlan<T> *pl, T *out, const T *terms, const T *key, size_t nterms,
                                   size_t nout, size_t batch, bool accumulate, hipStream_t st) {
    if (batch == 0 || nout == 0) return CNTT_OK;
    if (batch * std::max(nterms, nout) >= ((size_t)1 << 32)) return fail(CNTT_EINVAL, "batch too large for one launch");
    const size_t n = pl->n;
    if (nterms == 0) {  // empty sum
        if (!accumulate) HIP_TRY(hipMemsetAsync(out, 0, batch * nout * n * sizeof(T), st));
        return CNTT_OK;
    }
    DeviceTables<T> t;
    if (int rc = device_tables(pl, &t)) return rc;
    const int tcls = transform_class(pl);
    g_ext_split_wins = ext_split_wins(sizeof(T), pl->logn, tcls);
    const hipError_t e = launch_ext_ntt<T>(pl->logn, tcls, out, terms, key, alt_tables(tcls) ? t.fwd_fp : t.fwd,
                                           alt_tables(tcls) ? t.inv_fp : t.inv, pl->mp, (uint32_t)batch, (uint32_t)nterms,
                                           (uint32_t)nout, accumulate, st);
    if (e == hipSuccess) return CNTT_OK;
    if (e != hipErrorNotSupported) return fail(CNTT_EDEVICE, "fused mul_accumulate chain launch failed: %s", hipGetErrorString(e));
    (void)hipGetLastError();
    const size_t tw = batch * nterms * n, ow = batch * nout * n;
    T *scratch = nullptr;
    HIP_TRY(hipMallocAsync((void **)&scratch, (tw + (accumulate ? ow : 0)) * sizeof(T), st));
    T *tn = scratch, *acc = accumulate ? scratch + tw : out;
    int rc = CNTT_OK;
    do {
        if (hipMemcpyAsync(tn, terms, tw * sizeof(T), hipMemcpyDeviceToDevice, st) != hipSuccess) {
            rc = fail(CNTT_EDEVICE, "device copy failed");
            break;
        }
        if ((rc = ntt_device<T>(pl, tn, batch * nterms, false, st))) break;
        hipLaunchKernelGGL((ext_accumulate_kernel<T>), dim3(ew_grid(ow / (16 / sizeof(T)))), dim3(256), 0, st, acc, tn, key,
                           pl->mp, (uint32_t)pl->logn, (uint32_t)nterms, (uint32_t)nout, batch);
        if (hipGetLastError() != hipSuccess) {
            rc = fail(CNTT_EDEVICE, "ext_accumulate_kernel launch failed");
            break;
        }
        if ((rc = ntt_device<T>(pl, acc, batch * nout, true, st))) break;
        if (accumulate) rc = pointwise_device<T, PW_ADD>(pl, out, acc, nullptr, ow, st);
    } while (false);
    (void)hipFreeAsync(scratch, st);
    return rc;
}

template <class T>
static int external_product(const PrimePlan<T> *pl, T *out, const T *terms, const T *key, size_t nterms, size_t nout,
                            size_t batch, int accumulate, cntt_mem_t where, hipStream_t st) {
    if (!pl) return fail(CNTT_EINVAL, "plan is NULL");
    if (batch == 0 || nout == 0) return CNTT_OK;
    if (!out || (nterms && (!terms || !key))) return fail(CNTT_EINVAL, "NULL buffer");
    if (where == CNTT_MEM_DEVICE) return external_product_device<T>(pl, out, terms, key, nterms, nout, batch, accumulate != 0, st);
    const size_t n = pl->n, ob = batch * nout * n * sizeof(T), tb = batch * nterms * n * sizeof(T), kb = nterms * nout * n * sizeof(T);
    DevBuf dout, dt, dk;
    if (int rc = dout.alloc(ob)) return rc;
    if (int rc = dt.alloc(tb)) return rc;
    if (int rc = dk.alloc(kb)) return rc;
    if (accumulate) HIP_TRY(hipMemcpyAsync(dout.p, out, ob, hipMemcpyHostToDevice, st));
    if (tb) HIP_TRY(hipMemcpyAsync(dt.p, terms, tb, hipMemcpyHostToDevice, st));
    if (kb) HIP_TRY(hipMemcpyAsync(dk.p, key, kb, hipMemcpyHostToDevice, st));
    if (int rc = external_product_device<T>(pl, (T *)dout.p, (const T *)dt.p, (const T *)dk.p, nterms, nout, batch, accumulate != 0, st))
        return rc;
    HIP_TRY(hipMemcpyAsync(out, dout.p, ob, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    return CNTT_OK;
}

// op: 0 fwd, 1 inv, 2 mul_assign_normalize, 3 normalize, 4 mul_accumulate, 5 mul_ntt ; count = total elements
template <class T>
static int prime_op(const PrimePlan<T> *pl, int op, T *a, const T *b, const T *c, size_t count, size_t batch,
                    cntt_mem_t where, hipStream_t st) {
    if (!pl) return fail(CNTT_EINVAL, "plan is NULL");
    if (count == 0) return CNTT_OK;
    if (!a || ((op == 2 || op == 5) && !b) || (op == 4 && (!b || !c))) return fail(CNTT_EINVAL, "NULL buffer");
    auto run = [&](T *da, const T *db, const T *dc) -> int {
        switch (op) {
        case 0: return ntt_device<T>(pl, da, batch, false, st);
        case 1: return ntt_device<T>(pl, da, batch, true, st);
        case 2: return pointwise_device<T, PW_MUL_NORMALIZE>(pl, da, db, nullptr, count, st);
        case 3: return pointwise_device<T, PW_NORMALIZE>(pl, da, nullptr, nullptr, count, st);
        case 5: return mul_ntt_device<T>(pl, da, db, batch, st);
        default: return pointwise_device<T, PW_MUL_ACCUMULATE>(pl, da, db, dc, count, st);
        }
    };
    if (where == CNTT_MEM_DEVICE) return run(a, b, c);
    const size_t bytes = count * sizeof(T);
    DevBuf da, db, dc;
    if (int rc = da.alloc(bytes)) return rc;
    HIP_TRY(hipMemcpyAsync(da.p, a, bytes, hipMemcpyHostToDevice, st));
    if (op == 2 || op == 4 || op == 5) {
        if (int rc = db.alloc(bytes)) return rc;
        HIP_TRY(hipMemcpyAsync(db.p, b, bytes, hipMemcpyHostToDevice, st));
    }
    if (op == 4) {
        if (int rc = dc.alloc(bytes)) return rc;
        HIP_TRY(hipMemcpyAsync(dc.p, c, bytes, hipMemcpyHostToDevice, st));
    }
    if (int rc = run((T *)da.p, (const T *)db.p, (const T *)dc.p)) return rc;
    HIP_TRY(hipMemcpyAsync(a, da.p, bytes, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    return CNTT_OK;
}

template <class T> static int plan_info(const PrimePlan<T> *pl, cntt_plan_info_t *out) {
    if (!pl || !out) return fail(CNTT_EINVAL, "NULL argument");
    out->ntt_size = pl->n;
    out->modulus = pl->p;
    out->p_barrett = pl->p_barrett;
    out->big_q = pl->big_q;
    out->n_inv_mod_p = pl->n_inv;
    out->n_inv_mod_p_shoup = pl->n_inv_shoup;
    out->root = pl->root;
    out->has_shoup = pl->has_shoup ? 1 : 0;
    out->arith_class = (int32_t)transform_class(pl);
    return CNTT_OK;
}
template <class T> static int plan_table(const PrimePlan<T> *pl, cntt_table_t which, T *out, size_t len) {
    if (!pl || !out) return fail(CNTT_EINVAL, "NULL argument");
    if (len != pl->n) return fail(CNTT_ELEN, "len %zu != ntt_size %zu", len, pl->n);
    const std::vector<T> *src = nullptr;
    switch (which) {
    case CNTT_TWID: src = &pl->twid; break;
    case CNTT_TWID_SHOUP: src = &pl->twid_shoup; break;
    case CNTT_INV_TWID: src = &pl->inv_twid; break;
    case CNTT_INV_TWID_SHOUP: src = &pl->inv_twid_shoup; break;
    default: return fail(CNTT_EINVAL, "unknown table");
    }
    if (src->empty()) return fail(CNTT_NONE, "the plan has no Shoup tables (modulus >= 2^(B-1))");
    std::memcpy(out, src->data(), len * sizeof(T));
    return CNTT_OK;
}

// ---- C ABI: prime64 ---------------------------------------------------------------------------
#define CNTT_PRIME_API(BITS, T, PLAN)                                                                               \
    extern "C" int cntt_prime##BITS##_plan_new(size_t n, T p, PLAN **out) { return plan_new<T, PLAN>(n, p, out); }  \
    extern "C" PLAN *cntt_prime##BITS##_plan_clone(const PLAN *pl) {                                                \
        if (!pl) return nullptr;                                                                                    \
        return new (std::nothrow) PLAN(*pl); /* host tables copied, immutable device replicas shared */             \
    }                                                                                                               \
    extern "C" void cntt_prime##BITS##_plan_free(PLAN *pl) { delete pl; }                                           \
    extern "C" size_t cntt_prime##BITS##_ntt_size(const PLAN *pl) { return pl ? pl->n : 0; }                        \
    extern "C" T cntt_prime##BITS##_modulus(const PLAN *pl) { return pl ? pl->p : 0; }                              \
    extern "C" int cntt_prime##BITS##_plan_info(const PLAN *pl, cntt_plan_info_t *out) { return plan_info<T>(pl, out); } \
    extern "C" int cntt_prime##BITS##_plan_table(const PLAN *pl, cntt_table_t w, T *out, size_t len) {              \
        return plan_table<T>(pl, w, out, len);                                                                      \
    }                                                                                                               \
    extern "C" int cntt_prime##BITS##_fwd(const PLAN *pl, T *buf, size_t len) {                                     \
        if (!pl) return fail(CNTT_EINVAL, "plan is NULL");                                                          \
        if (len != pl->n) return fail(CNTT_ELEN, "assert_eq!(buf.len(), ntt_size): %zu != %zu", len, pl->n);        \
        return prime_op<T>(pl, 0, buf, nullptr, nullptr, len, 1, CNTT_MEM_HOST, nullptr);                           \
    }                                                                                                               \
    extern "C" int cntt_prime##BITS##_inv(const PLAN *pl, T *buf, size_t len) {                                     \
        if (!pl) return fail(CNTT_EINVAL, "plan is NULL");                                                          \
        if (len != pl->n) return fail(CNTT_ELEN, "assert_eq!(buf.len(), ntt_size): %zu != %zu", len, pl->n);        \
        return prime_op<T>(pl, 1, buf, nullptr, nullptr, len, 1, CNTT_MEM_HOST, nullptr);                           \
    }                                                                                                               \
    extern "C" int cntt_prime##BITS##_mul_assign_normalize(const PLAN *pl, T *lhs, size_t ll, const T *rhs, size_t rl) { \
        return prime_op<T>(pl, 2, lhs, rhs, nullptr, std::min(ll, rl), 0, CNTT_MEM_HOST, nullptr);                  \
    }                                                                                                               \
    extern "C" int cntt_prime##BITS##_normalize(const PLAN *pl, T *v, size_t len) {                                 \
        return prime_op<T>(pl, 3, v, nullptr, nullptr, len, 0, CNTT_MEM_HOST, nullptr);                             \
    }                                                                                                               \
    extern "C" int cntt_prime##BITS##_mul_accumulate(const PLAN *pl, T *acc, size_t al, const T *lhs, size_t ll,    \
                                                     const T *rhs, size_t rl) {                                     \
        return prime_op<T>(pl, 4, acc, lhs, rhs, std::min(al, std::min(ll, rl)), 0, CNTT_MEM_HOST, nullptr);        \
    }                                                                                                               \
    extern "C" int cntt_prime##BITS##_fwd_batch(const PLAN *pl, T *b, size_t batch, cntt_mem_t w, void *st) {       \
        return pl ? prime_op<T>(pl, 0, b, nullptr, nullptr, batch * pl->n, batch, w, (hipStream_t)st)               \
                  : fail(CNTT_EINVAL, "plan is NULL");                                                              \
    }                                                                                                               \
    extern "C" int cntt_prime##BITS##_inv_batch(const PLAN *pl, T *b, size_t batch, cntt_mem_t w, void *st) {       \
        return pl ? prime_op<T>(pl, 1, b, nullptr, nullptr, batch * pl->n, batch, w, (hipStream_t)st)               \
                  : fail(CNTT_EINVAL, "plan is NULL");                                                              \
    }                                                                                                               \
    extern "C" int cntt_prime##BITS##_mul_assign_normalize_batch(const PLAN *pl, T *l, const T *r, size_t batch,    \
                                                                 cntt_mem_t w, void *st) {                          \
        return pl ? prime_op<T>(pl, 2, l, r, nullptr, batch * pl->n, batch, w, (hipStream_t)st)                     \
                  : fail(CNTT_EINVAL, "plan is NULL");                                                              \
    }                                                                                                               \
    extern "C" int cntt_prime##BITS##_normalize_batch(const PLAN *pl, T *v, size_t batch, cntt_mem_t w, void *st) { \
        return pl ? prime_op<T>(pl, 3, v, nullptr, nullptr, batch * pl->n, batch, w, (hipStream_t)st)               \
                  : fail(CNTT_EINVAL, "plan is NULL");                                                              \
    }                                                                                                               \
    extern "C" int cntt_prime##BITS##_mul_accumulate_batch(const PLAN *pl, T *acc, const T *l, const T *r,          \
                                                           size_t batch, cntt_mem_t w, void *st) {                  \
        return pl ? prime_op<T>(pl, 4, acc, l, r, batch * pl->n, batch, w, (hipStream_t)st)                         \
                  : fail(CNTT_EINVAL, "plan is NULL");                                                              \
    }                                                                                                               \
    extern "C" int cntt_prime##BITS##_mul_ntt_batch(const PLAN *pl, T *l, const T *r, size_t batch, cntt_mem_t w,    \
                                                    void *st) {                                                     \
        return pl ? prime_op<T>(pl, 5, l, r, nullptr, batch * pl->n, batch, w, (hipStream_t)st)                     \
                  : fail(CNTT_EINVAL, "plan is NULL");                                                              \
    }                                                                                                               \
    extern "C" int cntt_prime##BITS##_external_product_batch(const PLAN *pl, T *out, const T *terms, const T *key_ntt,  \
                                                            size_t nterms, size_t nout, size_t batch, int accumulate,  \
                                                            cntt_mem_t where, void *stream) {                          \
        return external_product<T>(pl, out, terms, key_ntt, nterms, nout, batch, accumulate, where, (hipStream_t)stream); \
    }

CNTT_PRIME_API(64, uint64_t, cntt_plan64)
CNTT_PRIME_API(32, uint32_t, cntt_plan32)

// ---------------------------------------------------------------------------------------------
// fill
// ---------------------------------------------------------------------------------------------
extern "C" int cntt_fill_uniform_u64(uint64_t *dst, size_t count, uint64_t bound, uint64_t seed, void *st) {
    if (!dst && count) return fail(CNTT_EINVAL, "dst is NULL");
    if (!count) return CNTT_OK;
    hipLaunchKernelGGL((fill_uniform_kernel<uint64_t>), dim3(ew_grid(count)), dim3(256), 0, (hipStream_t)st, dst, count,
                       bound, seed);
    HIP_TRY(hipGetLastError());
    return CNTT_OK;
}
extern "C" int cntt_fill_uniform_u32(uint32_t *dst, size_t count, uint32_t bound, uint64_t seed, void *st) {
    if (!dst && count) return fail(CNTT_EINVAL, "dst is NULL");
    if (!count) return CNTT_OK;
    hipLaunchKernelGGL((fill_uniform_kernel<uint32_t>), dim3(ew_grid(count)), dim3(256), 0, (hipStream_t)st, dst, count,
                       bound, seed);
    HIP_TRY(hipGetLastError());
    return CNTT_OK;
}

// ---------------------------------------------------------------------------------------------
// native plans
// ---------------------------------------------------------------------------------------------
// src/lib.rs:453-462 and :601-606
static const uint32_t PRIMES32[10] = {1062862849u, 1063059457u, 1064697857u, 1065484289u, 1068236801u,
                                      1068433409u, 1068564481u, 1069219841u, 1071513601u, 1073479681u};
static const uint64_t PRIMES52[6] = {1125899881086977ull, 1125899885412353ull, 1125899886395393ull,
                                     1125899899174913ull, 1125899902124033ull, 1125899903107073ull};

struct NativeKindInfo {
    int nprimes, word, is52, binary, ngroups;
    int ga[5], gb[5];  // prime indices of each mixed-radix group (gb = -1: single prime)
};
static const NativeKindInfo NATIVE_KINDS[10] = {
    {3, 4, 0, 0, 3, {0, 1, 2, 0, 0}, {-1, -1, -1, -1, -1}},   // native32::Plan32           src/native32.rs:28-56
    {5, 8, 0, 0, 3, {0, 1, 3, 0, 0}, {-1, 2, 4, -1, -1}},     // native64::Plan32           src/native64.rs:91-141
    {10, 16, 0, 0, 5, {0, 2, 4, 6, 8}, {1, 3, 5, 7, 9}},      // native128::Plan32          src/native128.rs:20-118
    {2, 4, 0, 1, 2, {0, 1, 0, 0, 0}, {-1, -1, -1, -1, -1}},   // native_binary32::Plan32    src/native_binary32.rs:22-41
    {3, 8, 0, 1, 3, {0, 1, 2, 0, 0}, {-1, -1, -1, -1, -1}},   // native_binary64::Plan32    src/native_binary64.rs:33-61
    {5, 16, 0, 1, 3, {0, 1, 3, 0, 0}, {-1, 2, 4, -1, -1}},    // native_binary128::Plan32   src/native_binary128.rs:13-63
    {2, 4, 1, 0, 2, {0, 1, 0, 0, 0}, {-1, -1, -1, -1, -1}},   // native32::Plan52           src/native32.rs:223-253
    {3, 8, 1, 0, 3, {0, 1, 2, 0, 0}, {-1, -1, -1, -1, -1}},   // native64::Plan52           src/native64.rs:770-829
    {1, 4, 1, 1, 1, {0, 0, 0, 0, 0}, {-1, -1, -1, -1, -1}},   // native_binary32::Plan52    src/native_binary32.rs:111-125
    {2, 8, 1, 1, 2, {0, 1, 0, 0, 0}, {-1, -1, -1, -1, -1}},   // native_binary64::Plan52    src/native_binary64.rs:230-260
};

// One grow-only scratch area per (plan, device) for the composed native pipeline.  Calls from several threads or on
// several streams share it safely: the plan's mutex is held while a call enqueues its launches, and a call on another
// stream first waits for the event the previous user recorded (no wait, no record while a stream is being captured
// into a hipGraph: a graph replays against the buffer it captured, as cntt_native_reserve documents).
struct Workspace {
    void *base = nullptr;
    size_t bytes = 0;
    hipEvent_t last = nullptr;
    hipStream_t last_stream = nullptr;
};
struct NativeCache {
    std::mutex mu;
    std::map<int, Workspace> ws;
    ~NativeCache() {
        for (auto &kv : ws) {
            int cur = 0;
            if (hipGetDevice(&cur) != hipSuccess) continue;
            (void)hipSetDevice(kv.first);
            (void)hipFree(kv.second.base);
            if (kv.second.last) (void)hipEventDestroy(kv.second.last);
            (void)hipSetDevice(cur);
        }
    }
};

struct cntt_native {
    cntt_native_kind_t kind;
    NativeKindInfo info;
    size_t n = 0;
    std::vector<std::unique_ptr<cntt_plan32>> p32;
    std::vector<std::unique_ptr<cntt_plan64>> p64;
    CrtArgs crt{};
    // accumulating CRT of the whole-product kernel (native_fused.hpp, native_product_acc): constants, and the per-prime
    // parameters whose last-inverse-stage constants carry (M / P_i)^-1 / n
    AccArgs acc{};
    ModParams<uint32_t> mp_acc[10];
    bool has_acc = false;
    // Plan52 kinds (round 5): negacyclic_polymul returns the wrapping image of the EXACT integer product, which does not depend on the
    // primes it was computed with (native_fused.hpp, accumulating CRT) -- so it runs the whole-product kernel of the Plan32 kind with the
    // same words (30-bit primes: the faster arithmetic on this chip; the reference offers Plan52 for AVX-512 IFMA hosts).  fwd / inv of
    // the plan, whose 50-bit residues are visible, stay on its own primes.  Absent where the Plan32 kind does not exist (n < 32).
    std::unique_ptr<cntt_native> via32;
    std::shared_ptr<NativeCache> cache;
    size_t rbytes() const { return info.is52 ? 8 : 4; }
    uint64_t prime(int i) const { return info.is52 ? PRIMES52[i] : (uint64_t)PRIMES32[i]; }
};

static void build_crt_args(cntt_native *pl) {
    CrtArgs &A = pl->crt;
    const NativeKindInfo &I = pl->info;
    std::memset(&A, 0, sizeof A);
    A.k = I.nprimes;
    A.ngroups = I.ngroups;
    for (int i = 0; i < I.nprimes; ++i) A.prime[i] = pl->prime(i);
    u128 prefix = 1;
    std::vector<uint64_t> M((size_t)I.ngroups);
    for (int g = 0; g < I.ngroups; ++g) {
        const uint64_t pa = pl->prime(I.ga[g]);
        A.ga[g] = I.ga[g];
        A.gb[g] = I.gb[g];
        uint64_t m = pa;
        if (I.gb[g] >= 0) {
            const uint64_t pb = pl->prime(I.gb[g]);
            A.pair_inv[g] = host::powmod(pa % pb, pb - 2, pb);  // P_a^-1 mod P_b (src/lib.rs:536-561)
            A.pair_inv_shoup[g] = (uint32_t)((A.pair_inv[g] << 32) / pb);
            m = pa * pb;
        }
        M[(size_t)g] = m;
        A.M[g] = m;
        // the CRT kernels rely on ascending digit moduli (digits and group residues need no reduction modulo a later
        // modulus); true for every reference plan: primes ascend within primes32 / primes52 (src/lib.rs:447-652)
        if (g > 0 && !(M[(size_t)g - 1] < m)) std::abort();
        A.prefix_lo[g] = (uint64_t)prefix;
        A.prefix_hi[g] = (uint64_t)(prefix >> 64);
        if (g > 0) {
            // inv[g] = (M_0 ... M_{g-1})^-1 mod M[g]; M[g] is a prime or a product of two primes:
            // invert through Euler's theorem like src/lib.rs:541-551
            const uint64_t phi = (I.gb[g] >= 0) ? (pl->prime(I.ga[g]) - 1) * (pl->prime(I.gb[g]) - 1) : (m - 1);
            uint64_t pm = 1;  // true product M_0 ... M_{g-1} mod m (`prefix` itself wraps mod 2^128)
            for (int h = 0; h < g; ++h) pm = host::mulmod(pm, M[(size_t)h] % m, m);
            A.inv[g] = host::powmod(pm, phi - 1, m);
            A.inv_shoup[g] = (uint64_t)((((u128)A.inv[g]) << 64) / m);
            A.inv_shoup32[g] = (m >> 32) == 0 ? (uint32_t)((A.inv[g] << 32) / m) : 0u;
            for (int h = 0; h < g; ++h) {
                A.Mmod[g][h] = M[(size_t)h] % m;
                A.Mmod_shoup[g][h] = (uint64_t)((((u128)A.Mmod[g][h]) << 64) / m);
                A.Mmod_shoup32[g][h] = (m >> 32) == 0 ? (uint32_t)((A.Mmod[g][h] << 32) / m) : 0u;
            }
        }
        prefix *= (u128)m;  // wrapping mod 2^128, as src/lib.rs:592-595
    }
    A.prefix_lo[I.ngroups] = (uint64_t)prefix;
    A.prefix_hi[I.ngroups] = (uint64_t)(prefix >> 64);
}

// Plan32 kinds: M = P_0 ... P_{k-1}, M_i = M / P_i, y_i = M_i^-1 mod P_i (Euler, like src/lib.rs:541-551)
static void build_acc_args(cntt_native *pl) {
    const int k = pl->info.nprimes;
    if (pl->info.is52) return;
    AccArgs &A = pl->acc;
    std::memset(&A, 0, sizeof A);
    u128 m = 1;  // mod 2^128
    for (int i = 0; i < k; ++i) m *= (u128)PRIMES32[i];
    A.m_lo = (uint64_t)m;
    A.m_hi = (uint64_t)(m >> 64);
    for (int i = 0; i < k; ++i) {
        const uint64_t p = PRIMES32[i];
        u128 mi = 1;         // M_i mod 2^128
        uint64_t mi_p = 1;   // M_i mod P_i
        for (int h = 0; h < k; ++h) {
            if (h == i) continue;
            mi *= (u128)PRIMES32[h];
            mi_p = host::mulmod(mi_p, PRIMES32[h] % p, p);
        }
        A.c_lo[i] = (uint64_t)mi;
        A.c_hi[i] = (uint64_t)(mi >> 64);
        A.f[i] = (uint32_t)((((uint64_t)1) << (32 + ACC_FRAC_BITS)) / p);
        A.m60[i] = (uint32_t)((((uint64_t)1) << 60) / p);
        // the lazy split folds a word 32 bits at a time with t = hi c + lo < 2^58: needs c = 2^32 mod p < 2^26 - 1
        if (((((uint64_t)1) << 32) % p) >= (((uint64_t)1) << 26) - 1) return;
        // (M / P_i)^-1 times 2^32: the kernel's pointwise product is a Montgomery product (acc_mont_lazy) and leaves a factor 2^-32
        const uint64_t y = host::mulmod(host::powmod(mi_p, p - 2, p), (((uint64_t)1) << 32) % p, p);
        const cntt_plan32 *sub = pl->p32[(size_t)i].get();
        ModParams<uint32_t> mp = sub->mp;
        if (mp.cls != CLS_LAZY) return;
        mp.n_inv = (uint32_t)host::mulmod(mp.n_inv, y, p);
        mp.n_inv_shoup = shoup_of<uint32_t>(mp.n_inv, (uint32_t)p);
        mp.last_w = (uint32_t)host::mulmod(mp.last_w, y, p);
        mp.last_w_shoup = shoup_of<uint32_t>(mp.last_w, (uint32_t)p);
        pl->mp_acc[i] = mp;
    }
    pl->has_acc = true;
}
bool cntt::native_acc_enabled() {
    return debug_switch(DBG_NATIVE_ACC) != 0;   // A/B runs, parity tests
}

extern "C" int cntt_native_plan_new(cntt_native_kind_t kind, size_t n, cntt_native_t **out) {
    if (!out) return fail(CNTT_EINVAL, "out is NULL");
    *out = nullptr;
    if ((int)kind < 0 || (int)kind > 9) return fail(CNTT_EINVAL, "unknown native plan kind");
    std::unique_ptr<cntt_native> pl(new (std::nothrow) cntt_native());
    if (!pl) return fail(CNTT_ENOMEM, "out of memory");
    pl->kind = kind;
    pl->info = NATIVE_KINDS[kind];
    pl->n = n;
    for (int i = 0; i < pl->info.nprimes; ++i) {  // `?` propagation: src/native64.rs:933-942
        if (pl->info.is52) {
            cntt_plan64 *sub = nullptr;
            if (int rc = plan_new<uint64_t, cntt_plan64>(n, PRIMES52[i], &sub)) return rc;
            pl->p64.emplace_back(sub);
        } else {
            cntt_plan32 *sub = nullptr;
            if (int rc = plan_new<uint32_t, cntt_plan32>(n, PRIMES32[i], &sub)) return rc;
            pl->p32.emplace_back(sub);
        }
    }
    build_crt_args(pl.get());
    build_acc_args(pl.get());
    pl->cache = std::make_shared<NativeCache>();
    if (pl->info.is52) {
        static const cntt_native_kind_t SAME_WORDS[10] = {CNTT_NATIVE32_PLAN32, CNTT_NATIVE64_PLAN32, CNTT_NATIVE128_PLAN32,
                                                          CNTT_NATIVE_BINARY32_PLAN32, CNTT_NATIVE_BINARY64_PLAN32,
                                                          CNTT_NATIVE_BINARY128_PLAN32, CNTT_NATIVE32_PLAN32, CNTT_NATIVE64_PLAN32,
                                                          CNTT_NATIVE_BINARY32_PLAN32, CNTT_NATIVE_BINARY64_PLAN32};
        cntt_native_t *v = nullptr;
        if (cntt_native_plan_new(SAME_WORDS[kind], n, &v) == CNTT_OK) pl->via32.reset(v);   // None (n < 32 ...): composed path
    }
    *out = pl.release();
    return CNTT_OK;
}

extern "C" cntt_native_t *cntt_native_plan_clone(const cntt_native_t *pl) {
    if (!pl) return nullptr;
    cntt_native_t *out = nullptr;
    if (cntt_native_plan_new(pl->kind, pl->n, &out) != CNTT_OK) return nullptr;
    return out;
}
extern "C" void cntt_native_plan_free(cntt_native_t *pl) { delete pl; }
extern "C" size_t cntt_native_ntt_size(const cntt_native_t *pl) { return pl ? pl->n : 0; }
extern "C" int cntt_native_nprimes(const cntt_native_t *pl) { return pl ? pl->info.nprimes : 0; }
extern "C" int cntt_native_word_bytes(const cntt_native_t *pl) { return pl ? pl->info.word : 0; }
extern "C" int cntt_native_residue_bytes(const cntt_native_t *pl) { return pl ? (int)pl->rbytes() : 0; }
extern "C" const cntt_plan32_t *cntt_native_ntt32(const cntt_native_t *pl, int i) {
    if (!pl || pl->info.is52 || i < 0 || i >= pl->info.nprimes) return nullptr;
    return pl->p32[(size_t)i].get();
}
extern "C" const cntt_plan64_t *cntt_native_ntt64(const cntt_native_t *pl, int i) {
    if (!pl || !pl->info.is52 || i < 0 || i >= pl->info.nprimes) return nullptr;
    return pl->p64[(size_t)i].get();
}

template <class W, class R> static void launch_split(const void *value, const SplitArgs &A, size_t count, bool binary, hipStream_t st) {
    if (binary)
        hipLaunchKernelGGL((split_kernel<W, R, true>), dim3(ew_grid(count)), dim3(256), 0, st, (const W *)value, A, count);
    else
        hipLaunchKernelGGL((split_kernel<W, R, false>), dim3(ew_grid(count)), dim3(256), 0, st, (const W *)value, A, count);
}
struct W128 {
    uint64_t lo, hi;
};
static SplitArgs native_split_args(const cntt_native *pl, void *const *res) {
    SplitArgs A{};
    A.k = pl->info.nprimes;
    for (int i = 0; i < A.k; ++i) {
        const uint64_t p = pl->prime(i);
        A.res[i] = res ? res[i] : nullptr;
        A.prime[i] = p;
        if (pl->info.is52) {
            A.barrett[i] = (uint64_t)((((u128)1) << 64) / p);
        } else {
            const uint64_t c = (((uint64_t)1) << 32) % p;
            A.c[i] = (uint32_t)c;
            A.c_shoup[i] = (uint32_t)((c << 32) / p);
            A.one_shoup[i] = (uint32_t)((((uint64_t)1) << 32) / p);
        }
    }
    return A;
}
static int native_split_device(const cntt_native *pl, const void *value, void *const *res, size_t count, bool binary,
                               hipStream_t st) {
    const SplitArgs A = native_split_args(pl, res);
    // a u32 word is always below the 50-bit primes: src/native32.rs:447-452 copies it without `%`
    if (pl->info.is52) {
        if (pl->info.word == 4)
            launch_split<uint32_t, uint64_t>(value, A, count, true, st);
        else
            launch_split<uint64_t, uint64_t>(value, A, count, binary, st);
    } else {
        if (pl->info.word == 4)
            launch_split<uint32_t, uint32_t>(value, A, count, binary, st);
        else if (pl->info.word == 8)
            launch_split<uint64_t, uint32_t>(value, A, count, binary, st);
        else
            launch_split<W128, uint32_t>(value, A, count, binary, st);
    }
    HIP_TRY(hipGetLastError());
    return CNTT_OK;
}
template <class W, class R, int NG, uint32_t PAIRS>
static void launch_crt(void *value, const CrtArgs &A, size_t count, hipStream_t st) {
    hipLaunchKernelGGL((crt_kernel<W, R, NG, PAIRS>), dim3(ew_grid(count)), dim3(256), 0, st, (W *)value, A, count);
}
static int native_crt_device(const cntt_native *pl, void *value, void *const *res, size_t count, hipStream_t st) {
    CrtArgs A = pl->crt;
    for (int i = 0; i < A.k; ++i) A.res[i] = res[i];
    switch (pl->kind) {  // digit structure of each reference plan (NATIVE_KINDS)
    case CNTT_NATIVE32_PLAN32: launch_crt<uint32_t, uint32_t, 3, 0u>(value, A, count, st); break;
    case CNTT_NATIVE64_PLAN32: launch_crt<uint64_t, uint32_t, 3, 0b110u>(value, A, count, st); break;
    case CNTT_NATIVE128_PLAN32: launch_crt<W128, uint32_t, 5, 0b11111u>(value, A, count, st); break;
    case CNTT_NATIVE_BINARY32_PLAN32: launch_crt<uint32_t, uint32_t, 2, 0u>(value, A, count, st); break;
    case CNTT_NATIVE_BINARY64_PLAN32: launch_crt<uint64_t, uint32_t, 3, 0u>(value, A, count, st); break;
    case CNTT_NATIVE_BINARY128_PLAN32: launch_crt<W128, uint32_t, 3, 0b110u>(value, A, count, st); break;
    case CNTT_NATIVE32_PLAN52: launch_crt<uint32_t, uint64_t, 2, 0u>(value, A, count, st); break;
    case CNTT_NATIVE64_PLAN52: launch_crt<uint64_t, uint64_t, 3, 0u>(value, A, count, st); break;
    case CNTT_NATIVE_BINARY32_PLAN52: launch_crt<uint32_t, uint64_t, 1, 0u>(value, A, count, st); break;
    case CNTT_NATIVE_BINARY64_PLAN52: launch_crt<uint64_t, uint64_t, 2, 0u>(value, A, count, st); break;
    default: return fail(CNTT_EINVAL, "unknown native plan kind");
    }
    HIP_TRY(hipGetLastError());
    return CNTT_OK;
}
static int native_ntt_device(const cntt_native *pl, void *const *res, size_t batch, bool inv, hipStream_t st) {
    for (int i = 0; i < pl->info.nprimes; ++i) {
        int rc = pl->info.is52 ? ntt_device<uint64_t>(pl->p64[(size_t)i].get(), (uint64_t *)res[i], batch, inv, st)
                               : ntt_device<uint32_t>(pl->p32[(size_t)i].get(), (uint32_t *)res[i], batch, inv, st);
        if (rc) return rc;
    }
    return CNTT_OK;
}

// op: 0 fwd, 1 fwd_binary, 2 inv   (device pointers)
static int native_op_device(const cntt_native *pl, int op, void *value, void *const *res, size_t batch, hipStream_t st) {
    const size_t count = batch * pl->n;
    if (op == 2) {
        if (int rc = native_ntt_device(pl, res, batch, true, st)) return rc;
        return native_crt_device(pl, value, res, count, st);
    }
    if (int rc = native_split_device(pl, value, res, count, op == 1, st)) return rc;
    return native_ntt_device(pl, res, batch, false, st);
}

static int native_op(const cntt_native *pl, int op, void *value, void *const *res, size_t batch, cntt_mem_t where,
                     hipStream_t st) {
    if (!pl || !value || !res) return fail(CNTT_EINVAL, "NULL argument");
    if (op == 1 && !pl->info.binary) return fail(CNTT_EINVAL, "fwd_binary exists only on native_binary* plans");
    if (batch == 0) return CNTT_OK;
    const int k = pl->info.nprimes;
    for (int i = 0; i < k; ++i)
        if (!res[i]) return fail(CNTT_EINVAL, "NULL residue buffer");
    if (where == CNTT_MEM_DEVICE) return native_op_device(pl, op, value, res, batch, st);
    const size_t count = batch * pl->n, vbytes = count * (size_t)pl->info.word, rb = count * pl->rbytes();
    DevBuf dv;
    std::vector<DevBuf> dr((size_t)k);
    void *dres[10];
    if (int rc = dv.alloc(vbytes)) return rc;
    for (int i = 0; i < k; ++i) {
        if (int rc = dr[(size_t)i].alloc(rb)) return rc;
        dres[i] = dr[(size_t)i].p;
    }
    if (op == 2) {
        for (int i = 0; i < k; ++i) HIP_TRY(hipMemcpyAsync(dres[i], res[i], rb, hipMemcpyHostToDevice, st));
    } else {
        HIP_TRY(hipMemcpyAsync(dv.p, value, vbytes, hipMemcpyHostToDevice, st));
    }
    if (int rc = native_op_device(pl, op, dv.p, dres, batch, st)) return rc;
    // fwd writes the residues; inv writes the value AND the (inverse-transformed) residues: src/native64.rs:1010-1014
    for (int i = 0; i < k; ++i) HIP_TRY(hipMemcpyAsync(res[i], dres[i], rb, hipMemcpyDeviceToHost, st));
    if (op == 2) HIP_TRY(hipMemcpyAsync(value, dv.p, vbytes, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    return CNTT_OK;
}

extern "C" int cntt_native_fwd(const cntt_native_t *pl, const void *value, size_t len, void *const *res) {
    if (!pl) return fail(CNTT_EINVAL, "plan is NULL");
    if (len != pl->n) return fail(CNTT_ELEN, "assert_eq!(buf.len(), ntt_size): %zu != %zu", len, pl->n);
    return native_op(pl, 0, const_cast<void *>(value), res, 1, CNTT_MEM_HOST, nullptr);
}
extern "C" int cntt_native_fwd_binary(const cntt_native_t *pl, const void *value, size_t len, void *const *res) {
    if (!pl) return fail(CNTT_EINVAL, "plan is NULL");
    if (len != pl->n) return fail(CNTT_ELEN, "assert_eq!(buf.len(), ntt_size): %zu != %zu", len, pl->n);
    return native_op(pl, 1, const_cast<void *>(value), res, 1, CNTT_MEM_HOST, nullptr);
}
extern "C" int cntt_native_inv(const cntt_native_t *pl, void *value, size_t len, void *const *res) {
    if (!pl) return fail(CNTT_EINVAL, "plan is NULL");
    if (len != pl->n) return fail(CNTT_ELEN, "assert_eq!(buf.len(), ntt_size): %zu != %zu", len, pl->n);
    return native_op(pl, 2, value, res, 1, CNTT_MEM_HOST, nullptr);
}
extern "C" int cntt_native_fwd_batch(const cntt_native_t *pl, const void *value, void *const *res, size_t batch,
                                     cntt_mem_t where, void *st) {
    return native_op(pl, 0, const_cast<void *>(value), res, batch, where, (hipStream_t)st);
}
extern "C" int cntt_native_fwd_binary_batch(const cntt_native_t *pl, const void *value, void *const *res, size_t batch,
                                            cntt_mem_t where, void *st) {
    return native_op(pl, 1, const_cast<void *>(value), res, batch, where, (hipStream_t)st);
}
extern "C" int cntt_native_inv_batch(const cntt_native_t *pl, void *value, void *const *res, size_t batch,
                                     cntt_mem_t where, void *st) {
    return native_op(pl, 2, value, res, batch, where, (hipStream_t)st);
}

// bytes of workspace a polymul of `batch` products takes: the per-workgroup parking area of the large-n whole-product
// kernel (native_fused.hpp) where that kernel runs, otherwise both operands' residue arrays of the composed pipeline
static bool native_fusable(const cntt_native *pl, size_t batch) {
    if (batch == 0 || batch >= ((size_t)1 << 32) || pl->info.is52) return false;
    switch (pl->kind) {
    case CNTT_NATIVE32_PLAN32:
    case CNTT_NATIVE64_PLAN32:
    case CNTT_NATIVE128_PLAN32:
    case CNTT_NATIVE_BINARY32_PLAN32:
    case CNTT_NATIVE_BINARY64_PLAN32:
    case CNTT_NATIVE_BINARY128_PLAN32: return true;
    default: return false;
    }
}
static size_t native_park_bytes(const cntt_native *pl, size_t batch) {
    if (!native_fusable(pl, batch)) return 0;
    return sizeof(uint32_t) * native_fused_scratch_words((int)pl->kind, pl->p32[0]->logn, pl->info.nprimes, device_num_cus(),
                                                        (uint32_t)batch);
}
static size_t native_workspace_bytes(const cntt_native *pl, size_t batch) {
    const size_t park = native_park_bytes(pl, batch);
    if (park) return park;
    // the register-resident whole-product kernel (accumulating CRT) and the LDS-parked one (32 <= n <= 4096, every fused kind but
    // native128) need no workspace at all
    if (native_fusable(pl, batch) && pl->has_acc && native_fused_acc((int)pl->kind, pl->p32[0]->logn) && native_acc_enabled()) return 0;
    if (native_fusable(pl, batch) && pl->p32[0]->logn >= 5 && pl->p32[0]->logn <= 12 && pl->kind != CNTT_NATIVE128_PLAN32) return 0;
    return 2 * (size_t)pl->info.nprimes * batch * pl->n * pl->rbytes();
}
// Plan52 kinds: does negacyclic_polymul run the Plan32 whole-product kernel of the same words?  (Measured, ns per product, through it /
// composed on the 50-bit primes, profiles/r05_plan52_via32.txt: native64 n = 4096 102 / 187, native32 n = 1024 12 / 28.5,
// native_binary64 n = 16384 384 / 572 ...; the one shape where the composed pipeline wins is native64 at n = 32768: 1816 / 1769.)
static bool plan52_via32(const cntt_native *pl) {
    if (!pl->via32 || debug_switch(DBG_PLAN52_VIA32) == 0) return false;
    return !(pl->kind == CNTT_NATIVE64_PLAN52 && pl->n >= 32768);
}
// caller holds pl->cache->mu
static int native_workspace(const cntt_native *pl, size_t need, Workspace **out) {
    int dev = 0;
    HIP_TRY(hipGetDevice(&dev));
    Workspace &w = pl->cache->ws[dev];
    if (w.bytes < need) {
        if (w.base) {
            HIP_TRY(hipDeviceSynchronize());  // the old workspace may still be in use by enqueued work
            (void)hipFree(w.base);
            w.base = nullptr;
            w.bytes = 0;
        }
        if (hipMalloc(&w.base, need) != hipSuccess) {
            w.base = nullptr;
            return fail(CNTT_ENOMEM, "hipMalloc(%zu) for the native workspace failed", need);
        }
        w.bytes = need;
    }
    *out = &w;
    return CNTT_OK;
}
extern "C" int cntt_native_reserve(const cntt_native_t *pl, size_t batch) {
    if (!pl) return fail(CNTT_EINVAL, "plan is NULL");
    if (plan52_via32(pl)) return cntt_native_reserve(pl->via32.get(), batch);
    std::lock_guard<std::mutex> lk(pl->cache->mu);
    Workspace *w = nullptr;
    return native_workspace(pl, native_workspace_bytes(pl, batch), &w);
}

// negacyclic_polymul on device memory: src/native64.rs:1042-1069 batched
// whole product in one kernel (native_fused.hpp) for the Plan32 kinds, 32 <= n <= 16384
template <int KIND>
static hipError_t native_fused_try(const cntt_native *pl, void *prod, const void *lhs, const void *rhs, size_t batch,
                                   uint32_t *park, hipStream_t st, int *rc_out) {
    constexpr int KP = NativeShape<KIND>::KP;
    FusedTables<KP> F{}, Facc{};
    for (int i = 0; i < KP; ++i) {
        DeviceTables<uint32_t> t;
        if (int rc = device_tables(pl->p32[(size_t)i].get(), &t)) {
            *rc_out = rc;
            return hipErrorUnknown;
        }
        F.twf[i] = Facc.twf[i] = t.fwd;
        F.twi[i] = Facc.twi[i] = t.inv;
        F.P[i] = pl->p32[(size_t)i]->mp;
        Facc.P[i] = pl->mp_acc[i];
    }
    const SplitArgs S = native_split_args(pl, nullptr);
    return launch_native_fused<KIND>(pl->p32[0]->logn, prod, lhs, rhs, &F, S, pl->crt, (uint32_t)batch, park, st,
                                     pl->has_acc ? &pl->acc : nullptr, pl->has_acc ? &Facc : nullptr);
}
// CNTT_OK: enqueued; FUSED_NONE: no whole-product kernel for this plan / size (the caller composes)
static constexpr int FUSED_NONE = -1;
static int native_fused_device(const cntt_native *pl, void *prod, const void *lhs, const void *rhs, size_t batch,
                               uint32_t *park, hipStream_t st) {
    int rc = CNTT_OK;
    hipError_t e = hipErrorNotSupported;
    switch (pl->kind) {
    case CNTT_NATIVE32_PLAN32: e = native_fused_try<0>(pl, prod, lhs, rhs, batch, park, st, &rc); break;
    case CNTT_NATIVE64_PLAN32: e = native_fused_try<1>(pl, prod, lhs, rhs, batch, park, st, &rc); break;
    case CNTT_NATIVE128_PLAN32: e = native_fused_try<2>(pl, prod, lhs, rhs, batch, park, st, &rc); break;
    case CNTT_NATIVE_BINARY32_PLAN32: e = native_fused_try<3>(pl, prod, lhs, rhs, batch, park, st, &rc); break;
    case CNTT_NATIVE_BINARY64_PLAN32: e = native_fused_try<4>(pl, prod, lhs, rhs, batch, park, st, &rc); break;
    case CNTT_NATIVE_BINARY128_PLAN32: e = native_fused_try<5>(pl, prod, lhs, rhs, batch, park, st, &rc); break;
    default: break;
    }
    if (rc != CNTT_OK) return rc;
    if (e == hipSuccess) return CNTT_OK;
    if (e != hipErrorNotSupported) return fail(CNTT_EDEVICE, "fused polymul launch failed: %s", hipGetErrorString(e));
    (void)hipGetLastError();
    return FUSED_NONE;
}

static int native_polymul_device(const cntt_native *pl, void *prod, const void *lhs, const void *rhs, size_t batch,
                                 hipStream_t st) {
    if (plan52_via32(pl)) return native_polymul_device(pl->via32.get(), prod, lhs, rhs, batch, st);
    const size_t park = native_park_bytes(pl, batch);
    if (park == 0 && native_fusable(pl, batch)) {
        const int rc = native_fused_device(pl, prod, lhs, rhs, batch, nullptr, st);
        if (rc != FUSED_NONE) return rc;
    }
    std::lock_guard<std::mutex> lk(pl->cache->mu);  // held while this call's launches are enqueued
    Workspace *ws = nullptr;
    if (int rc = native_workspace(pl, native_workspace_bytes(pl, batch), &ws)) return rc;
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    (void)hipStreamIsCapturing(st, &cap);
    const bool capturing = cap != hipStreamCaptureStatusNone;
    if (!capturing && ws->last && ws->last_stream != st) HIP_TRY(hipStreamWaitEvent(st, ws->last, 0));
    struct Release {  // records "done with the workspace" on every exit path
        Workspace *w;
        hipStream_t st;
        bool on;
        ~Release() {
            if (!on) return;
            if (!w->last && hipEventCreateWithFlags(&w->last, hipEventDisableTiming) != hipSuccess) w->last = nullptr;
            if (w->last) (void)hipEventRecord(w->last, st);
            w->last_stream = st;
        }
    } release{ws, st, !capturing};
    void *base = ws->base;
    if (park) {  // n = 8192 / 16384: the persistent whole-product kernel, its workgroups parking residue tiles in the workspace
        const int rc = native_fused_device(pl, prod, lhs, rhs, batch, (uint32_t *)base, st);
        return rc == FUSED_NONE ? fail(CNTT_EDEVICE, "no whole-product kernel for n = %zu", pl->n) : rc;
    }
    const int k = pl->info.nprimes;
    const size_t count = batch * pl->n, rb = count * pl->rbytes();
    void *L[10], *R[10];
    for (int i = 0; i < k; ++i) {
        L[i] = (char *)base + (size_t)i * rb;
        R[i] = (char *)base + (size_t)(k + i) * rb;
    }
    // rhs: split + forward transforms; lhs: split, then per prime the fused
    // inv(mul_assign_normalize(fwd(lhs_i), rhs_i^)) (one kernel for n <= 2048, three launches otherwise); CRT.
    if (int rc = native_op_device(pl, pl->info.binary ? 1 : 0, const_cast<void *>(rhs), R, batch, st)) return rc;
    if (int rc = native_split_device(pl, lhs, L, count, false, st)) return rc;
    for (int i = 0; i < k; ++i) {
        int rc = pl->info.is52 ? mul_ntt_device<uint64_t>(pl->p64[(size_t)i].get(), (uint64_t *)L[i],
                                                         (const uint64_t *)R[i], batch, st)
                               : mul_ntt_device<uint32_t>(pl->p32[(size_t)i].get(), (uint32_t *)L[i],
                                                         (const uint32_t *)R[i], batch, st);
        if (rc) return rc;
    }
    return native_crt_device(pl, prod, L, count, st);
}

extern "C" int cntt_native_negacyclic_polymul_batch(const cntt_native_t *pl, void *prod, const void *lhs,
                                                    const void *rhs, size_t batch, cntt_mem_t where, void *stream) {
    if (!pl || !prod || !lhs || !rhs) return fail(CNTT_EINVAL, "NULL argument");
    if (batch == 0) return CNTT_OK;
    hipStream_t st = (hipStream_t)stream;
    if (where == CNTT_MEM_DEVICE) return native_polymul_device(pl, prod, lhs, rhs, batch, st);
    const size_t vbytes = batch * pl->n * (size_t)pl->info.word;
    DevBuf dp, dl, dr;
    if (int rc = dp.alloc(vbytes)) return rc;
    if (int rc = dl.alloc(vbytes)) return rc;
    if (int rc = dr.alloc(vbytes)) return rc;
    HIP_TRY(hipMemcpyAsync(dl.p, lhs, vbytes, hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemcpyAsync(dr.p, rhs, vbytes, hipMemcpyHostToDevice, st));
    if (int rc = native_polymul_device(pl, dp.p, dl.p, dr.p, batch, st)) return rc;
    HIP_TRY(hipMemcpyAsync(prod, dp.p, vbytes, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    return CNTT_OK;
}
extern "C" int cntt_native_negacyclic_polymul(const cntt_native_t *pl, void *prod, size_t pn, const void *lhs, size_t ln,
                                              const void *rhs, size_t rn) {
    if (!pl) return fail(CNTT_EINVAL, "plan is NULL");
    // assert_eq!(n, lhs.len()); assert_eq!(n, rhs.len()) then the inner fwd asserts ntt_size: src/native64.rs:1043-1045
    if (pn != ln || pn != rn) return fail(CNTT_ELEN, "prod/lhs/rhs lengths differ");
    if (pn != pl->n) return fail(CNTT_ELEN, "assert_eq!(buf.len(), ntt_size): %zu != %zu", pn, pl->n);
    return cntt_native_negacyclic_polymul_batch(pl, prod, lhs, rhs, 1, CNTT_MEM_HOST, nullptr);
}

// ---------------------------------------------------------------------------------------------
// product::Plan  (src/product.rs:139-967)
// ---------------------------------------------------------------------------------------------
struct cntt_product {
    size_t n = 0;
    uint64_t modulus = 0;
    std::vector<uint64_t> primes;  // ascending, 1s removed
    std::vector<std::shared_ptr<cntt_plan32>> p32;
    std::vector<std::shared_ptr<cntt_plan64>> p64;
    std::vector<uint64_t> modular_inverses;  // src/product.rs:207-229
    ProductArgs args{};
    size_t len32() const { return (n / 2) * p32.size(); }          // ntt_domain_len_u32 src/product.rs:261-263
    size_t domain_len() const { return len32() + n * p64.size(); } // ntt_domain_len     src/product.rs:268-270
};

// Plan::try_new src/product.rs:153-247
extern "C" int cntt_product_plan_new(size_t n, uint64_t modulus, const uint64_t *factors, size_t nfactors,
                                     cntt_product_t **out) {
    if (!out) return fail(CNTT_EINVAL, "out is NULL");
    *out = nullptr;
    if (nfactors && !factors) return fail(CNTT_EINVAL, "factors is NULL");
    if (n % 2 != 0) return CNTT_NONE;
    std::vector<uint64_t> primes(factors, factors + nfactors);
    std::sort(primes.begin(), primes.end());
    uint64_t prev = 0;
    for (uint64_t f : primes) {  // zero or repeated factor: src/product.rs:163-169
        if (f == prev) return CNTT_NONE;
        prev = f;
    }
    primes.erase(primes.begin(), std::find_if(primes.begin(), primes.end(), [](uint64_t f) { return f != 1; }));
    uint64_t prod = 1;
    for (uint64_t f : primes) {  // checked_mul: src/product.rs:173-177
        const u128 w = (u128)prod * f;
        if (w >> 64) return CNTT_NONE;
        prod = (uint64_t)w;
    }
    if (prod != modulus) return CNTT_NONE;
    // distinct primes = 1 mod 2n >= 65 whose product fits u64: never more than 7; anything longer has a
    // non-prime factor and try_new of that factor would return None anyway
    if (primes.size() >= (size_t)PRODUCT_MAX_PRIMES) return CNTT_NONE;
    std::unique_ptr<cntt_product> pl(new (std::nothrow) cntt_product());
    if (!pl) return fail(CNTT_ENOMEM, "out of memory");
    pl->n = n;
    pl->modulus = modulus;
    pl->primes = primes;
    for (uint64_t f : primes) {
        if (f < ((uint64_t)1 << 32)) {
            cntt_plan32 *sub = nullptr;
            if (int rc = plan_new<uint32_t, cntt_plan32>(n, (uint32_t)f, &sub)) return rc;
            pl->p32.emplace_back(sub);
        } else {
            cntt_plan64 *sub = nullptr;
            if (int rc = plan_new<uint64_t, cntt_plan64>(n, f, &sub)) return rc;
            pl->p64.emplace_back(sub);
        }
    }
    ProductArgs &A = pl->args;
    A.n32 = (int)pl->p32.size();
    A.n64 = (int)pl->p64.size();
    A.modulus = modulus;
    for (size_t j = 0; j < primes.size(); ++j) {
        A.prime[j] = primes[j];
        A.barrett[j] = (uint64_t)((((u128)1) << 64) / primes[j]);
        for (size_t i = 0; i < j; ++i) {  // every factor is prime here, so Fermat gives the Euclid inverse of :22-64
            const uint64_t inv = host::powmod(primes[i] % primes[j], primes[j] - 2, primes[j]);
            pl->modular_inverses.push_back(inv);
            A.inv[j * (j - 1) / 2 + i] = inv;
            A.inv_shoup[j * (j - 1) / 2 + i] = (uint64_t)((((u128)inv) << 64) / primes[j]);
        }
    }
    *out = pl.release();
    return CNTT_OK;
}
extern "C" cntt_product_t *cntt_product_plan_clone(const cntt_product_t *pl) {
    return pl ? new (std::nothrow) cntt_product(*pl) : nullptr;  // prime plans are immutable and shared
}
extern "C" void cntt_product_plan_free(cntt_product_t *pl) { delete pl; }
extern "C" size_t cntt_product_ntt_size(const cntt_product_t *pl) { return pl ? pl->n : 0; }
extern "C" uint64_t cntt_product_modulus(const cntt_product_t *pl) { return pl ? pl->modulus : 0; }
extern "C" size_t cntt_product_ntt_domain_len(const cntt_product_t *pl) { return pl ? pl->domain_len() : 0; }
extern "C" int cntt_product_nprimes32(const cntt_product_t *pl) { return pl ? (int)pl->p32.size() : 0; }
extern "C" int cntt_product_nprimes64(const cntt_product_t *pl) { return pl ? (int)pl->p64.size() : 0; }
extern "C" uint64_t cntt_product_prime(const cntt_product_t *pl, int i) {
    return (pl && i >= 0 && (size_t)i < pl->primes.size()) ? pl->primes[(size_t)i] : 0;
}
extern "C" const cntt_plan32_t *cntt_product_ntt32(const cntt_product_t *pl, int i) {
    return (pl && i >= 0 && (size_t)i < pl->p32.size()) ? pl->p32[(size_t)i].get() : nullptr;
}
extern "C" const cntt_plan64_t *cntt_product_ntt64(const cntt_product_t *pl, int i) {
    return (pl && i >= 0 && (size_t)i < pl->p64.size()) ? pl->p64[(size_t)i].get() : nullptr;
}
extern "C" int cntt_product_modular_inverses(const cntt_product_t *pl, uint64_t *out, size_t len) {
    if (!pl || (!out && len)) return fail(CNTT_EINVAL, "NULL argument");
    if (len != pl->modular_inverses.size()) return fail(CNTT_ELEN, "expected %zu inverses", pl->modular_inverses.size());
    std::copy(pl->modular_inverses.begin(), pl->modular_inverses.end(), out);
    return CNTT_OK;
}

// plane-major device views of a batched NTT-domain buffer (see aux_kernels.hpp)
struct ProductView {
    uint32_t *r32;
    uint64_t *r64;
};
static ProductView product_view(const cntt_product *pl, uint64_t *ntt, size_t batch) {
    return {reinterpret_cast<uint32_t *>(ntt), ntt + pl->len32() * batch};
}
static int product_ntt_device(const cntt_product *pl, ProductView v, size_t batch, bool inv, hipStream_t st) {
    const size_t count = batch * pl->n;
    for (size_t k = 0; k < pl->p32.size(); ++k)
        if (int rc = ntt_device<uint32_t>(pl->p32[k].get(), v.r32 + k * count, batch, inv, st)) return rc;
    for (size_t k = 0; k < pl->p64.size(); ++k)
        if (int rc = ntt_device<uint64_t>(pl->p64[k].get(), v.r64 + k * count, batch, inv, st)) return rc;
    return CNTT_OK;
}

// residue split of `batch` polynomials into the plane-major view v (the first half of Plan::fwd, src/product.rs:282-355)
static int product_split_device(const cntt_product *pl, ProductView v, const uint64_t *standard, size_t batch, bool bounded,
                                uint64_t bound, hipStream_t st) {
    const size_t count = batch * pl->n, k = pl->primes.size();
    if (count == 0 || k == 0) return CNTT_OK;
    ProductArgs A = pl->args;
    A.bound = bound;
    const dim3 grid(ew_grid(count / 2)), block(256);
    if (k == 1)
        hipLaunchKernelGGL((product_split_kernel<2>), grid, block, 0, st, v.r32, v.r64, standard, A, count);
    else if (A.n32 == 2 && A.n64 == 0 && bounded && bound < A.prime[0] && bound < A.prime[1])
        hipLaunchKernelGGL((product_split_kernel<1>), grid, block, 0, st, v.r32, v.r64, standard, A, count);
    else
        hipLaunchKernelGGL((product_split_kernel<0>), grid, block, 0, st, v.r32, v.r64, standard, A, count);
    HIP_TRY(hipGetLastError());
    return CNTT_OK;
}

// u32x2 plans whose primes share an arithmetic class: split + both forward transforms, or both inverse transforms +
// Garner, in one kernel (product_fused.hpp).  Returns hipErrorNotSupported when the plan / size is not covered.
static hipError_t product_fused2_try(const cntt_product *pl, bool inv, uint64_t *standard, uint32_t *res32, size_t batch,
                                     bool flag, hipStream_t st, int *rc_out) {
    if (pl->p32.size() != 2 || !pl->p64.empty() || batch == 0 || batch >= ((size_t)1 << 32)) return hipErrorNotSupported;
    // Round 3: with the element-wise kernels on uncapped grids the composed forward (split kernel + two batched transforms) is
    // 8 % faster than the fused forward kernel (N = 2048, 32768 polynomials: 0.497 vs 0.542 ms) -- the fused one reads its
    // twiddles from L2 at three wavefronts per SIMD, the batched transforms from an LDS image -- while the fused inverse
    // (two transforms + Garner, no residue round trip) still wins (Replace 0.445 vs 0.522 ms; Accumulate 0.599 vs 0.582: a
    // tie).  cntt_debug_set("product_fused", 0 / 1) forces neither / both for A/B timing; results are identical (tests/test_product.py).
    const int force = debug_switch(DBG_PRODUCT_FUSED);
    if (force == 0 || (force < 0 && !inv)) return hipErrorNotSupported;
    const cntt_plan32 *q0 = pl->p32[0].get(), *q1 = pl->p32[1].get();
    const int cls = transform_class(q0);  // both primes above 2^31 (the reference's fast-path shape): CLS_FPW
    if (cls != transform_class(q1)) return hipErrorNotSupported;
    ProductFusedTables F{};
    for (int i = 0; i < 2; ++i) {
        DeviceTables<uint32_t> t;
        if (int rc = device_tables(pl->p32[(size_t)i].get(), &t)) {
            *rc_out = rc;
            return hipErrorUnknown;
        }
        F.twf[i] = alt_tables(cls) ? t.fwd_fp : t.fwd;
        F.twi[i] = alt_tables(cls) ? t.inv_fp : t.inv;
        F.P[i] = pl->p32[(size_t)i]->mp;
    }
    return launch_product_fused2(q0->logn, cls, inv, standard, res32, &F, pl->args, (uint32_t)batch, flag, st);
}

// Plan::fwd src/product.rs:273-357  (device pointers)
static int product_fwd_device(const cntt_product *pl, uint64_t *ntt, const uint64_t *standard, size_t batch,
                              bool bounded, uint64_t bound, hipStream_t st) {
    if (batch == 0 || pl->primes.empty()) return CNTT_OK;
    const ProductView v = product_view(pl, ntt, batch);
    {
        int rc = CNTT_OK;
        const bool fast = bounded && pl->p32.size() == 2 && bound < pl->args.prime[0] && bound < pl->args.prime[1];
        const hipError_t e = product_fused2_try(pl, false, const_cast<uint64_t *>(standard), v.r32, batch, fast, st, &rc);
        if (rc != CNTT_OK) return rc;
        if (e == hipSuccess) return CNTT_OK;
        if (e != hipErrorNotSupported) return fail(CNTT_EDEVICE, "fused product forward launch failed: %s", hipGetErrorString(e));
        (void)hipGetLastError();
    }
    if (int rc = product_split_device(pl, v, standard, batch, bounded, bound, st)) return rc;
    return product_ntt_device(pl, v, batch, false, st);
}

template <int K>
static void launch_product_crt(uint64_t *standard, ProductView v, const ProductArgs &A, size_t count, int acc,
                               hipStream_t st) {
    const dim3 grid(ew_grid(count / 2)), block(256);
    if (acc == 0) hipLaunchKernelGGL((product_crt_kernel<K, 0>), grid, block, 0, st, standard, v.r32, v.r64, A, count);
    else if (acc == 1) hipLaunchKernelGGL((product_crt_kernel<K, 1>), grid, block, 0, st, standard, v.r32, v.r64, A, count);
    else hipLaunchKernelGGL((product_crt_kernel<K, 2>), grid, block, 0, st, standard, v.r32, v.r64, A, count);
}

// Garner recombination of `batch` polynomials from the plane-major view v (the second half of Plan::inv, src/product.rs:386-879)
static int product_crt_device(const cntt_product *pl, uint64_t *standard, ProductView v, size_t batch, bool accumulate,
                              hipStream_t st) {
    const size_t count = batch * pl->n, k = pl->primes.size();
    if (count == 0) return CNTT_OK;
    const int acc = !accumulate ? 0 : (k == 1 && pl->p32.size() == 1 ? 2 : 1);
    switch (k) {
    case 1: launch_product_crt<1>(standard, v, pl->args, count, acc, st); break;
    case 2: launch_product_crt<2>(standard, v, pl->args, count, acc, st); break;
    case 3: launch_product_crt<3>(standard, v, pl->args, count, acc, st); break;
    case 4: launch_product_crt<4>(standard, v, pl->args, count, acc, st); break;
    case 5: launch_product_crt<5>(standard, v, pl->args, count, acc, st); break;
    case 6: launch_product_crt<6>(standard, v, pl->args, count, acc, st); break;
    default: launch_product_crt<7>(standard, v, pl->args, count, acc, st); break;
    }
    HIP_TRY(hipGetLastError());
    return CNTT_OK;
}

// Plan::inv src/product.rs:360-879  (device pointers)
static int product_inv_device(const cntt_product *pl, uint64_t *standard, uint64_t *ntt, size_t batch, bool accumulate,
                              hipStream_t st) {
    const size_t count = batch * pl->n;
    if (count == 0) return CNTT_OK;
    if (pl->primes.empty()) {  // src/product.rs:378-384
        if (!accumulate) HIP_TRY(hipMemsetAsync(standard, 0, count * 8, st));
        return CNTT_OK;
    }
    const ProductView v = product_view(pl, ntt, batch);
    {
        int rc = CNTT_OK;
        const hipError_t e = product_fused2_try(pl, true, standard, v.r32, batch, accumulate, st, &rc);
        if (rc != CNTT_OK) return rc;
        if (e == hipSuccess) return CNTT_OK;
        if (e != hipErrorNotSupported) return fail(CNTT_EDEVICE, "fused product inverse launch failed: %s", hipGetErrorString(e));
        (void)hipGetLastError();
    }
    if (int rc = product_ntt_device(pl, v, batch, true, st)) return rc;
    return product_crt_device(pl, standard, v, batch, accumulate, st);
}

// The external-product step of the reference's caller at the product::Plan level (device pointers):
//     for j { plan.fwd(t_j, terms[b][j], fwd_mode); for o { plan.mul_accumulate(acc_o, t_j, key[j][o]) } }
//     for o { plan.inv(out[b][o], acc_o, inv_mode) }
// = residue split of all terms, one fused mul_accumulate chain per prime plane, one Garner recombination.
static int product_external_product_device(const cntt_product *pl, uint64_t *out, const uint64_t *terms, const uint64_t *key,
                                           size_t nterms, size_t nout, size_t batch, bool bounded, uint64_t bound,
                                           bool accumulate, hipStream_t st) {
    if (batch == 0 || nout == 0) return CNTT_OK;
    const size_t n = pl->n, dl = pl->domain_len();
    if (pl->primes.empty() || nterms == 0) {
        if (!accumulate) HIP_TRY(hipMemsetAsync(out, 0, batch * nout * n * 8, st));
        return CNTT_OK;
    }
    if (pl->primes.size() == 1 && pl->p64.size() == 1)
        // u64x1 plan: fwd copies (src/product.rs:282-286) and inv copies / add_mod_u64s (:386-398), so the per-prime
        // chain reads `terms` and writes `out` directly -- no residue buffers at all
        return external_product_device<uint64_t>(pl->p64[0].get(), out, terms, key, nterms, nout, batch, accumulate, st);
    const size_t tpolys = batch * nterms, opolys = batch * nout, kpolys = nterms * nout;
    uint64_t *scratch = nullptr;
    HIP_TRY(hipMallocAsync((void **)&scratch, (tpolys + opolys) * dl * 8, st));
    uint64_t *tres = scratch, *ores = scratch + tpolys * dl;
    const ProductView tv = product_view(pl, tres, tpolys), ov = product_view(pl, ores, opolys);
    const ProductView kv = product_view(pl, const_cast<uint64_t *>(key), kpolys);
    int rc = product_split_device(pl, tv, terms, tpolys, bounded, bound, st);
    for (size_t i = 0; i < pl->p32.size() && rc == CNTT_OK; ++i)
        rc = external_product_device<uint32_t>(pl->p32[i].get(), ov.r32 + i * opolys * n, tv.r32 + i * tpolys * n,
                                               kv.r32 + i * kpolys * n, nterms, nout, batch, false, st);
    for (size_t i = 0; i < pl->p64.size() && rc == CNTT_OK; ++i)
        rc = external_product_device<uint64_t>(pl->p64[i].get(), ov.r64 + i * opolys * n, tv.r64 + i * tpolys * n,
                                               kv.r64 + i * kpolys * n, nterms, nout, batch, false, st);
    if (rc == CNTT_OK) rc = product_crt_device(pl, out, ov, opolys, accumulate, st);
    (void)hipFreeAsync(scratch, st);
    return rc;
}

// op 2 mul_assign_normalize, 3 normalize, 4 mul_accumulate, per prime on the plane-major layout: src/product.rs:885-966
static int product_pointwise_device(const cntt_product *pl, int op, uint64_t *a, const uint64_t *b, const uint64_t *c,
                                    size_t batch, hipStream_t st) {
    const size_t count = batch * pl->n;
    if (count == 0) return CNTT_OK;
    const size_t off64 = pl->len32() * batch;
    for (size_t k = 0; k < pl->p32.size() + pl->p64.size(); ++k) {
        int rc;
        if (k < pl->p32.size()) {
            const cntt_plan32 *sub = pl->p32[k].get();
            uint32_t *pa = reinterpret_cast<uint32_t *>(a) + k * count;
            const uint32_t *pb = b ? reinterpret_cast<const uint32_t *>(b) + k * count : nullptr;
            const uint32_t *pc = c ? reinterpret_cast<const uint32_t *>(c) + k * count : nullptr;
            rc = op == 2   ? pointwise_device<uint32_t, PW_MUL_NORMALIZE>(sub, pa, pb, nullptr, count, st)
                 : op == 3 ? pointwise_device<uint32_t, PW_NORMALIZE>(sub, pa, nullptr, nullptr, count, st)
                           : pointwise_device<uint32_t, PW_MUL_ACCUMULATE>(sub, pa, pb, pc, count, st);
        } else {
            const size_t o = off64 + (k - pl->p32.size()) * count;
            const cntt_plan64 *sub = pl->p64[k - pl->p32.size()].get();
            rc = op == 2   ? pointwise_device<uint64_t, PW_MUL_NORMALIZE>(sub, a + o, b + o, nullptr, count, st)
                 : op == 3 ? pointwise_device<uint64_t, PW_NORMALIZE>(sub, a + o, nullptr, nullptr, count, st)
                           : pointwise_device<uint64_t, PW_MUL_ACCUMULATE>(sub, a + o, b + o, c + o, count, st);
        }
        if (rc) return rc;
    }
    return CNTT_OK;
}

// op: 0 fwd (a = ntt out, b = standard in), 1 inv (a = standard, b = ntt, both written),
//     2 mul_assign_normalize (a lhs, b rhs), 3 normalize (a), 4 mul_accumulate (a acc, b lhs, c rhs)
static int product_op(const cntt_product *pl, int op, uint64_t *a, uint64_t *b, const uint64_t *c, size_t batch, int mode,
                      uint64_t bound, cntt_mem_t where, hipStream_t st) {
    if (!pl) return fail(CNTT_EINVAL, "plan is NULL");
    if (batch == 0) return CNTT_OK;
    if (batch * pl->n >= ((size_t)1 << 40)) return fail(CNTT_EINVAL, "batch too large");
    const size_t std_words = batch * pl->n, dom_words = batch * pl->domain_len();
    const size_t aw = op == 1 ? std_words : dom_words, bw = op == 0 ? std_words : dom_words;
    if ((!a && aw) || (op != 3 && !b && bw) || (op == 4 && !c && dom_words)) return fail(CNTT_EINVAL, "NULL buffer");
    auto run = [&](uint64_t *da, uint64_t *db, const uint64_t *dc) -> int {
        switch (op) {
        case 0: return product_fwd_device(pl, da, db, batch, mode != 0, bound, st);
        case 1: return product_inv_device(pl, da, db, batch, mode != 0, st);
        default: return product_pointwise_device(pl, op, da, db, dc, batch, st);
        }
    };
    if (where == CNTT_MEM_DEVICE) return run(a, b, c);
    DevBuf da, db, dc;
    if (int rc = da.alloc(aw * 8)) return rc;
    if (op != 0 && !(op == 1 && mode == 0)) HIP_TRY(hipMemcpyAsync(da.p, a, aw * 8, hipMemcpyHostToDevice, st));
    if (op != 3) {
        if (int rc = db.alloc(bw * 8)) return rc;
        HIP_TRY(hipMemcpyAsync(db.p, b, bw * 8, hipMemcpyHostToDevice, st));
    }
    if (op == 4) {
        if (int rc = dc.alloc(dom_words * 8)) return rc;
        HIP_TRY(hipMemcpyAsync(dc.p, c, dom_words * 8, hipMemcpyHostToDevice, st));
    }
    if (int rc = run((uint64_t *)da.p, (uint64_t *)db.p, (const uint64_t *)dc.p)) return rc;
    if (!(op == 1 && mode != 0 && pl->primes.empty()) && aw) HIP_TRY(hipMemcpyAsync(a, da.p, aw * 8, hipMemcpyDeviceToHost, st));
    // inv leaves the inverse-transformed residues in the caller's ntt buffer: src/product.rs:368-373
    if (op == 1 && bw) HIP_TRY(hipMemcpyAsync(b, db.p, bw * 8, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    return CNTT_OK;
}

#define PRODUCT_LEN(have, want, what)                                                                          \
    if ((have) != (want)) return fail(CNTT_ELEN, "assert_eq!(" what "): %zu != %zu", (size_t)(have), (size_t)(want))

extern "C" int cntt_product_fwd(const cntt_product_t *pl, uint64_t *ntt, size_t ntt_len, const uint64_t *standard,
                                size_t standard_len, cntt_fwd_mode_t mode, uint64_t bound) {
    if (!pl) return fail(CNTT_EINVAL, "plan is NULL");
    PRODUCT_LEN(standard_len, pl->n, "standard.len(), ntt_size");
    PRODUCT_LEN(ntt_len, pl->domain_len(), "ntt.len(), ntt_domain_len");
    return product_op(pl, 0, ntt, const_cast<uint64_t *>(standard), nullptr, 1, (int)mode, bound, CNTT_MEM_HOST, nullptr);
}
extern "C" int cntt_product_inv(const cntt_product_t *pl, uint64_t *standard, size_t standard_len, uint64_t *ntt,
                                size_t ntt_len, cntt_inv_mode_t mode) {
    if (!pl) return fail(CNTT_EINVAL, "plan is NULL");
    PRODUCT_LEN(standard_len, pl->n, "standard.len(), ntt_size");
    PRODUCT_LEN(ntt_len, pl->domain_len(), "ntt.len(), ntt_domain_len");
    return product_op(pl, 1, standard, ntt, nullptr, 1, (int)mode, 0, CNTT_MEM_HOST, nullptr);
}
extern "C" int cntt_product_mul_assign_normalize(const cntt_product_t *pl, uint64_t *lhs, size_t lhs_len,
                                                 const uint64_t *rhs, size_t rhs_len) {
    if (!pl) return fail(CNTT_EINVAL, "plan is NULL");
    PRODUCT_LEN(lhs_len, pl->domain_len(), "lhs.len(), ntt_domain_len");
    PRODUCT_LEN(rhs_len, pl->domain_len(), "rhs.len(), ntt_domain_len");
    return product_op(pl, 2, lhs, const_cast<uint64_t *>(rhs), nullptr, 1, 0, 0, CNTT_MEM_HOST, nullptr);
}
extern "C" int cntt_product_normalize(const cntt_product_t *pl, uint64_t *values, size_t len) {
    if (!pl) return fail(CNTT_EINVAL, "plan is NULL");
    PRODUCT_LEN(len, pl->domain_len(), "values.len(), ntt_domain_len");
    return product_op(pl, 3, values, nullptr, nullptr, 1, 0, 0, CNTT_MEM_HOST, nullptr);
}
extern "C" int cntt_product_mul_accumulate(const cntt_product_t *pl, uint64_t *acc, size_t acc_len, const uint64_t *lhs,
                                           size_t lhs_len, const uint64_t *rhs, size_t rhs_len) {
    if (!pl) return fail(CNTT_EINVAL, "plan is NULL");
    PRODUCT_LEN(lhs_len, pl->domain_len(), "lhs.len(), ntt_domain_len");
    PRODUCT_LEN(rhs_len, pl->domain_len(), "rhs.len(), ntt_domain_len");
    PRODUCT_LEN(acc_len, pl->domain_len(), "acc.len(), ntt_domain_len");
    return product_op(pl, 4, acc, const_cast<uint64_t *>(lhs), rhs, 1, 0, 0, CNTT_MEM_HOST, nullptr);
}
extern "C" int cntt_product_fwd_batch(const cntt_product_t *pl, uint64_t *ntt, const uint64_t *standard, size_t batch,
                                      cntt_fwd_mode_t mode, uint64_t bound, cntt_mem_t where, void *stream) {
    return product_op(pl, 0, ntt, const_cast<uint64_t *>(standard), nullptr, batch, (int)mode, bound, where, (hipStream_t)stream);
}
extern "C" int cntt_product_inv_batch(const cntt_product_t *pl, uint64_t *standard, uint64_t *ntt, size_t batch,
                                      cntt_inv_mode_t mode, cntt_mem_t where, void *stream) {
    return product_op(pl, 1, standard, ntt, nullptr, batch, (int)mode, 0, where, (hipStream_t)stream);
}
extern "C" int cntt_product_mul_assign_normalize_batch(const cntt_product_t *pl, uint64_t *lhs, const uint64_t *rhs,
                                                       size_t batch, cntt_mem_t where, void *stream) {
    return product_op(pl, 2, lhs, const_cast<uint64_t *>(rhs), nullptr, batch, 0, 0, where, (hipStream_t)stream);
}
extern "C" int cntt_product_normalize_batch(const cntt_product_t *pl, uint64_t *values, size_t batch, cntt_mem_t where,
                                            void *stream) {
    return product_op(pl, 3, values, nullptr, nullptr, batch, 0, 0, where, (hipStream_t)stream);
}
extern "C" int cntt_product_mul_accumulate_batch(const cntt_product_t *pl, uint64_t *acc, const uint64_t *lhs,
                                                 const uint64_t *rhs, size_t batch, cntt_mem_t where, void *stream) {
    return product_op(pl, 4, acc, const_cast<uint64_t *>(lhs), rhs, batch, 0, 0, where, (hipStream_t)stream);
}

extern "C" int cntt_product_external_product_batch(const cntt_product_t *pl, uint64_t *out, const uint64_t *terms,
                                                   const uint64_t *key_ntt, size_t nterms, size_t nout, size_t batch,
                                                   cntt_fwd_mode_t fwd_mode, uint64_t bound, cntt_inv_mode_t inv_mode,
                                                   cntt_mem_t where, void *stream) {
    if (!pl) return fail(CNTT_EINVAL, "plan is NULL");
    if (batch == 0 || nout == 0) return CNTT_OK;
    const size_t n = pl->n, dl = pl->domain_len();
    if (!out || (nterms && (!terms || (dl && !key_ntt)))) return fail(CNTT_EINVAL, "NULL buffer");
    if (batch * std::max(nterms, nout) * n >= ((size_t)1 << 40)) return fail(CNTT_EINVAL, "batch too large");
    hipStream_t st = (hipStream_t)stream;
    const bool bounded = fwd_mode == CNTT_FWD_BOUNDED, accumulate = inv_mode == CNTT_INV_ACCUMULATE;
    if (where == CNTT_MEM_DEVICE)
        return product_external_product_device(pl, out, terms, key_ntt, nterms, nout, batch, bounded, bound, accumulate, st);
    const size_t ob = batch * nout * n * 8, tb = batch * nterms * n * 8, kb = nterms * nout * dl * 8;
    DevBuf dout, dt, dk;
    if (int rc = dout.alloc(ob)) return rc;
    if (int rc = dt.alloc(tb)) return rc;
    if (int rc = dk.alloc(kb)) return rc;
    if (accumulate) HIP_TRY(hipMemcpyAsync(dout.p, out, ob, hipMemcpyHostToDevice, st));
    if (tb) HIP_TRY(hipMemcpyAsync(dt.p, terms, tb, hipMemcpyHostToDevice, st));
    if (kb) HIP_TRY(hipMemcpyAsync(dk.p, key_ntt, kb, hipMemcpyHostToDevice, st));
    if (int rc = product_external_product_device(pl, (uint64_t *)dout.p, (const uint64_t *)dt.p, (const uint64_t *)dk.p, nterms,
                                                 nout, batch, bounded, bound, accumulate, st))
        return rc;
    HIP_TRY(hipMemcpyAsync(out, dout.p, ob, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    return CNTT_OK;
}
