// cntt.hpp -- header-only C++17 mirror of the concrete-ntt API over the C ABI of libcntt_hip.so (cntt.h).
//
// Same module / type / method names and the same error behaviour as the crate (src/lib.rs:88-110):
//   cntt::prime32::Plan, cntt::prime64::Plan            try_new -> std::optional, fwd / inv / mul_assign_normalize /
//                                                       normalize / mul_accumulate on caller-owned slices
//   cntt::native32 / native64 / native128::Plan32, native_binary32 / 64 / 128::Plan32, and the Plan52 variants
//   cntt::product::Plan, product::FwdMode, product::InvMode
// Where the reference returns None this returns std::nullopt; where it panics (length asserts, modulus <= 1) this
// throws cntt::Panic; HIP failures -- including "no GPU": there is no CPU path -- throw cntt::DeviceError.
// The *_batch methods are the device-resident batched extension (see cntt.h for layouts).
// INPUT CONTRACT (cntt.h): coefficients handed to fwd / inv / the pointwise calls are canonical, 0 <= x < modulus.  Words >= modulus are
// neither rejected nor reduced: the call completes, the affected outputs are unspecified residues (as in the reference, whose back ends
// disagree with each other outside that range).
#pragma once
#include <cstddef>
#include <cstdint>
#include <optional>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#include "cntt.h"

namespace cntt {

struct Panic : std::logic_error {
    using std::logic_error::logic_error;
};
struct DeviceError : std::runtime_error {
    using std::runtime_error::runtime_error;
};

namespace detail {
inline void check(int rc) {
    if (rc == CNTT_OK) return;
    const std::string msg = cntt_last_error();
    if (rc == CNTT_EINVAL || rc == CNTT_ELEN) throw Panic(msg);
    if (rc == CNTT_ENOMEM) throw std::bad_alloc();
    throw DeviceError(msg);
}
// `try_new`: CNTT_NONE -> nullopt, other failures throw
template <class H, class F> std::optional<H *> try_handle(F &&make) {
    H *h = nullptr;
    const int rc = make(&h);
    if (rc == CNTT_NONE) return std::nullopt;
    check(rc);
    return h;
}

template <class T> struct PrimeApi;
#define CNTT_PRIME_TRAITS(BITS, T)                                                                                   \
    template <> struct PrimeApi<T> {                                                                                 \
        using handle = cntt_plan##BITS##_t;                                                                          \
        static int plan_new(size_t n, T p, handle **o) { return cntt_prime##BITS##_plan_new(n, p, o); }               \
        static handle *clone(const handle *h) { return cntt_prime##BITS##_plan_clone(h); }                            \
        static void free(handle *h) { cntt_prime##BITS##_plan_free(h); }                                              \
        static size_t ntt_size(const handle *h) { return cntt_prime##BITS##_ntt_size(h); }                            \
        static T modulus(const handle *h) { return cntt_prime##BITS##_modulus(h); }                                   \
        static int fwd(const handle *h, T *b, size_t n) { return cntt_prime##BITS##_fwd(h, b, n); }                   \
        static int inv(const handle *h, T *b, size_t n) { return cntt_prime##BITS##_inv(h, b, n); }                   \
        static int mul_assign_normalize(const handle *h, T *l, size_t ln, const T *r, size_t rn) {                    \
            return cntt_prime##BITS##_mul_assign_normalize(h, l, ln, r, rn);                                          \
        }                                                                                                            \
        static int normalize(const handle *h, T *v, size_t n) { return cntt_prime##BITS##_normalize(h, v, n); }       \
        static int mul_accumulate(const handle *h, T *a, size_t an, const T *l, size_t ln, const T *r, size_t rn) {   \
            return cntt_prime##BITS##_mul_accumulate(h, a, an, l, ln, r, rn);                                         \
        }                                                                                                            \
        static int fwd_batch(const handle *h, T *b, size_t k, cntt_mem_t w, void *s) {                                \
            return cntt_prime##BITS##_fwd_batch(h, b, k, w, s);                                                       \
        }                                                                                                            \
        static int inv_batch(const handle *h, T *b, size_t k, cntt_mem_t w, void *s) {                                \
            return cntt_prime##BITS##_inv_batch(h, b, k, w, s);                                                       \
        }                                                                                                            \
        static int mul_assign_normalize_batch(const handle *h, T *l, const T *r, size_t k, cntt_mem_t w, void *s) {   \
            return cntt_prime##BITS##_mul_assign_normalize_batch(h, l, r, k, w, s);                                   \
        }                                                                                                            \
        static int mul_ntt_batch(const handle *h, T *l, const T *r, size_t k, cntt_mem_t w, void *s) {                \
            return cntt_prime##BITS##_mul_ntt_batch(h, l, r, k, w, s);                                                \
        }                                                                                                            \
        static int external_product_batch(const handle *h, T *o, const T *t, const T *key, size_t j, size_t no,      \
                                          size_t k, int acc, cntt_mem_t w, void *s) {                                 \
            return cntt_prime##BITS##_external_product_batch(h, o, t, key, j, no, k, acc, w, s);                      \
        }                                                                                                            \
    };
CNTT_PRIME_TRAITS(32, uint32_t)
CNTT_PRIME_TRAITS(64, uint64_t)
#undef CNTT_PRIME_TRAITS

// prime32::Plan / prime64::Plan (src/prime32.rs:601-927, src/prime64.rs:221-1129)
template <class T> class PrimePlan {
    using A = PrimeApi<T>;
    typename A::handle *h_ = nullptr;
    bool owned_ = true;

  public:
    explicit PrimePlan(typename A::handle *h, bool owned = true) : h_(h), owned_(owned) {}
    PrimePlan(const PrimePlan &o) : h_(A::clone(o.h_)) {}  // #[derive(Clone)]
    PrimePlan(PrimePlan &&o) noexcept : h_(std::exchange(o.h_, nullptr)), owned_(o.owned_) {}
    PrimePlan &operator=(PrimePlan o) noexcept {
        std::swap(h_, o.h_);
        std::swap(owned_, o.owned_);
        return *this;
    }
    ~PrimePlan() {
        if (h_ && owned_) A::free(h_);
    }
    static std::optional<PrimePlan> try_new(size_t polynomial_size, T modulus) {
        auto h = try_handle<typename A::handle>([&](auto **o) { return A::plan_new(polynomial_size, modulus, o); });
        if (!h) return std::nullopt;
        return PrimePlan(*h);
    }
    size_t ntt_size() const { return A::ntt_size(h_); }
    T modulus() const { return A::modulus(h_); }
    void fwd(T *buf, size_t len) const { check(A::fwd(h_, buf, len)); }
    void inv(T *buf, size_t len) const { check(A::inv(h_, buf, len)); }
    void fwd(std::vector<T> &buf) const { fwd(buf.data(), buf.size()); }
    void inv(std::vector<T> &buf) const { inv(buf.data(), buf.size()); }
    void mul_assign_normalize(T *lhs, size_t ln, const T *rhs, size_t rn) const { check(A::mul_assign_normalize(h_, lhs, ln, rhs, rn)); }
    void mul_assign_normalize(std::vector<T> &lhs, const std::vector<T> &rhs) const {
        mul_assign_normalize(lhs.data(), lhs.size(), rhs.data(), rhs.size());
    }
    void normalize(T *v, size_t n) const { check(A::normalize(h_, v, n)); }
    void mul_accumulate(T *acc, size_t an, const T *lhs, size_t ln, const T *rhs, size_t rn) const {
        check(A::mul_accumulate(h_, acc, an, lhs, ln, rhs, rn));
    }
    // batched extension: `batch` polynomials back to back in device (or host) memory, enqueued on `stream`
    void fwd_batch(T *bufs, size_t batch, cntt_mem_t where = CNTT_MEM_DEVICE, void *stream = nullptr) const {
        check(A::fwd_batch(h_, bufs, batch, where, stream));
    }
    void inv_batch(T *bufs, size_t batch, cntt_mem_t where = CNTT_MEM_DEVICE, void *stream = nullptr) const {
        check(A::inv_batch(h_, bufs, batch, where, stream));
    }
    void mul_assign_normalize_batch(T *lhs, const T *rhs, size_t batch, cntt_mem_t where = CNTT_MEM_DEVICE, void *stream = nullptr) const {
        check(A::mul_assign_normalize_batch(h_, lhs, rhs, batch, where, stream));
    }
    void mul_ntt_batch(T *lhs, const T *rhs_ntt, size_t batch, cntt_mem_t where = CNTT_MEM_DEVICE, void *stream = nullptr) const {
        check(A::mul_ntt_batch(h_, lhs, rhs_ntt, batch, where, stream));
    }
    void external_product_batch(T *out, const T *terms, const T *key_ntt, size_t nterms, size_t nout, size_t batch,
                                bool accumulate = false, cntt_mem_t where = CNTT_MEM_DEVICE, void *stream = nullptr) const {
        check(A::external_product_batch(h_, out, terms, key_ntt, nterms, nout, batch, accumulate ? 1 : 0, where, stream));
    }
    const typename A::handle *handle() const { return h_; }
};

// native / native_binary plans: W = coefficient word (uint32_t, uint64_t, or u128 as two uint64_t), R = residue word
template <cntt_native_kind_t KIND, class R, int NPRIMES, int WORD_BYTES> class NativePlan {
    cntt_native_t *h_ = nullptr;

  public:
    explicit NativePlan(cntt_native_t *h) : h_(h) {}
    NativePlan(const NativePlan &o) : h_(cntt_native_plan_clone(o.h_)) {}
    NativePlan(NativePlan &&o) noexcept : h_(std::exchange(o.h_, nullptr)) {}
    NativePlan &operator=(NativePlan o) noexcept {
        std::swap(h_, o.h_);
        return *this;
    }
    ~NativePlan() {
        if (h_) cntt_native_plan_free(h_);
    }
    static std::optional<NativePlan> try_new(size_t polynomial_size) {
        auto h = try_handle<cntt_native_t>([&](auto **o) { return cntt_native_plan_new(KIND, polynomial_size, o); });
        if (!h) return std::nullopt;
        return NativePlan(*h);
    }
    size_t ntt_size() const { return cntt_native_ntt_size(h_); }
    static constexpr int nprimes() { return NPRIMES; }
    // ntt_0() .. ntt_k(): borrowed prime plan
    PrimePlan<R> ntt(int i) const {
        if constexpr (sizeof(R) == 4)
            return PrimePlan<R>(const_cast<cntt_plan32_t *>(cntt_native_ntt32(h_, i)), false);
        else
            return PrimePlan<R>(const_cast<cntt_plan64_t *>(cntt_native_ntt64(h_, i)), false);
    }
    // value: ntt_size words of WORD_BYTES bytes; residues: NPRIMES buffers of ntt_size R words
    void fwd(const void *value, size_t len, R *const (&residues)[NPRIMES]) const {
        check(cntt_native_fwd(h_, value, len, reinterpret_cast<void *const *>(residues)));
    }
    void fwd_binary(const void *value, size_t len, R *const (&residues)[NPRIMES]) const {
        check(cntt_native_fwd_binary(h_, value, len, reinterpret_cast<void *const *>(residues)));
    }
    void inv(void *value, size_t len, R *const (&residues)[NPRIMES]) const {
        check(cntt_native_inv(h_, value, len, reinterpret_cast<void *const *>(residues)));
    }
    void negacyclic_polymul(void *prod, size_t pn, const void *lhs, size_t ln, const void *rhs, size_t rn) const {
        check(cntt_native_negacyclic_polymul(h_, prod, pn, lhs, ln, rhs, rn));
    }
    void negacyclic_polymul_batch(void *prod, const void *lhs, const void *rhs, size_t batch,
                                  cntt_mem_t where = CNTT_MEM_DEVICE, void *stream = nullptr) const {
        check(cntt_native_negacyclic_polymul_batch(h_, prod, lhs, rhs, batch, where, stream));
    }
    void reserve(size_t batch) const { check(cntt_native_reserve(h_, batch)); }
};
}  // namespace detail

// Contiguous shard [begin, end) of `rank` when a batch of independent polynomials is split over `world` devices (cntt_shard_bounds;
// SURVEY 8e): the caller drives the devices -- one thread + stream per device around a shared plan, examples/multi_device.cpp.
inline std::pair<size_t, size_t> shard_bounds(size_t batch, int world, int rank) {
    size_t b = 0, e = 0;
    detail::check(cntt_shard_bounds(batch, world, rank, &b, &e));
    return {b, e};
}
// TESTING ONLY (include/cntt.h): kernel-selection switches for A/B timing and device-vs-device parity tests; results never change.
inline void debug_set(const char *key, int value) { detail::check(cntt_debug_set(key, value)); }
inline int debug_get(const char *key) {
    int v = 0;
    detail::check(cntt_debug_get(key, &v));
    return v;
}


namespace prime32 {
using Plan = detail::PrimePlan<uint32_t>;
}
namespace prime64 {
using Plan = detail::PrimePlan<uint64_t>;
struct Solinas {
    static constexpr uint64_t P = 0xFFFFFFFF00000001ull;  // src/prime64/generic_solinas.rs:35-40
};
}  // namespace prime64
namespace native32 {
using Plan32 = detail::NativePlan<CNTT_NATIVE32_PLAN32, uint32_t, 3, 4>;
using Plan52 = detail::NativePlan<CNTT_NATIVE32_PLAN52, uint64_t, 2, 4>;
}  // namespace native32
namespace native64 {
using Plan32 = detail::NativePlan<CNTT_NATIVE64_PLAN32, uint32_t, 5, 8>;
using Plan52 = detail::NativePlan<CNTT_NATIVE64_PLAN52, uint64_t, 3, 8>;
}  // namespace native64
namespace native128 {
using Plan32 = detail::NativePlan<CNTT_NATIVE128_PLAN32, uint32_t, 10, 16>;
}
namespace native_binary32 {
using Plan32 = detail::NativePlan<CNTT_NATIVE_BINARY32_PLAN32, uint32_t, 2, 4>;
using Plan52 = detail::NativePlan<CNTT_NATIVE_BINARY32_PLAN52, uint64_t, 1, 4>;
}  // namespace native_binary32
namespace native_binary64 {
using Plan32 = detail::NativePlan<CNTT_NATIVE_BINARY64_PLAN32, uint32_t, 3, 8>;
using Plan52 = detail::NativePlan<CNTT_NATIVE_BINARY64_PLAN52, uint64_t, 2, 8>;
}  // namespace native_binary64
namespace native_binary128 {
using Plan32 = detail::NativePlan<CNTT_NATIVE_BINARY128_PLAN32, uint32_t, 5, 16>;
}

// product::Plan (src/product.rs:139-967)
namespace product {
struct FwdMode {  // enum FwdMode { Generic, Bounded(u64) }  src/product.rs:124-129
    bool bounded = false;
    uint64_t bound = 0;
    static FwdMode Generic() { return {}; }
    static FwdMode Bounded(uint64_t b) { return {true, b}; }
};
enum class InvMode { Replace = CNTT_INV_REPLACE, Accumulate = CNTT_INV_ACCUMULATE };  // src/product.rs:131-136

class Plan {
    cntt_product_t *h_ = nullptr;

  public:
    explicit Plan(cntt_product_t *h) : h_(h) {}
    Plan(const Plan &o) : h_(cntt_product_plan_clone(o.h_)) {}
    Plan(Plan &&o) noexcept : h_(std::exchange(o.h_, nullptr)) {}
    Plan &operator=(Plan o) noexcept {
        std::swap(h_, o.h_);
        return *this;
    }
    ~Plan() {
        if (h_) cntt_product_plan_free(h_);
    }
    static std::optional<Plan> try_new(size_t polynomial_size, uint64_t modulus, const std::vector<uint64_t> &factors) {
        auto h = detail::try_handle<cntt_product_t>(
            [&](auto **o) { return cntt_product_plan_new(polynomial_size, modulus, factors.data(), factors.size(), o); });
        if (!h) return std::nullopt;
        return Plan(*h);
    }
    size_t ntt_size() const { return cntt_product_ntt_size(h_); }
    uint64_t modulus() const { return cntt_product_modulus(h_); }
    size_t ntt_domain_len() const { return cntt_product_ntt_domain_len(h_); }
    void fwd(std::vector<uint64_t> &ntt, const std::vector<uint64_t> &standard, FwdMode mode) const {
        detail::check(cntt_product_fwd(h_, ntt.data(), ntt.size(), standard.data(), standard.size(),
                                       mode.bounded ? CNTT_FWD_BOUNDED : CNTT_FWD_GENERIC, mode.bound));
    }
    void inv(std::vector<uint64_t> &standard, std::vector<uint64_t> &ntt, InvMode mode) const {
        detail::check(cntt_product_inv(h_, standard.data(), standard.size(), ntt.data(), ntt.size(), (cntt_inv_mode_t)mode));
    }
    void mul_assign_normalize(std::vector<uint64_t> &lhs, const std::vector<uint64_t> &rhs) const {
        detail::check(cntt_product_mul_assign_normalize(h_, lhs.data(), lhs.size(), rhs.data(), rhs.size()));
    }
    void normalize(std::vector<uint64_t> &values) const { detail::check(cntt_product_normalize(h_, values.data(), values.size())); }
    void mul_accumulate(std::vector<uint64_t> &acc, const std::vector<uint64_t> &lhs, const std::vector<uint64_t> &rhs) const {
        detail::check(cntt_product_mul_accumulate(h_, acc.data(), acc.size(), lhs.data(), lhs.size(), rhs.data(), rhs.size()));
    }
    // batched extension (plane-major NTT domain, see cntt.h)
    void fwd_batch(uint64_t *ntt, const uint64_t *standard, size_t batch, FwdMode mode, cntt_mem_t where = CNTT_MEM_DEVICE,
                   void *stream = nullptr) const {
        detail::check(cntt_product_fwd_batch(h_, ntt, standard, batch, mode.bounded ? CNTT_FWD_BOUNDED : CNTT_FWD_GENERIC,
                                             mode.bound, where, stream));
    }
    void inv_batch(uint64_t *standard, uint64_t *ntt, size_t batch, InvMode mode, cntt_mem_t where = CNTT_MEM_DEVICE,
                   void *stream = nullptr) const {
        detail::check(cntt_product_inv_batch(h_, standard, ntt, batch, (cntt_inv_mode_t)mode, where, stream));
    }
    void external_product_batch(uint64_t *out, const uint64_t *terms, const uint64_t *key_ntt, size_t nterms, size_t nout,
                                size_t batch, FwdMode fmode, InvMode imode, cntt_mem_t where = CNTT_MEM_DEVICE,
                                void *stream = nullptr) const {
        detail::check(cntt_product_external_product_batch(h_, out, terms, key_ntt, nterms, nout, batch,
                                                          fmode.bounded ? CNTT_FWD_BOUNDED : CNTT_FWD_GENERIC, fmode.bound,
                                                          (cntt_inv_mode_t)imode, where, stream));
    }
};
}  // namespace product

}  // namespace cntt
