// Host-side number theory for plan construction (the product's own implementation; the test oracle
// under oracle/ is a separate code base and is never linked here).
//
// What has to agree with the reference bit for bit is the *choice* of the primitive 2N-th root
// (src/roots.rs:68-91: start from p-1 and take log2(2N)-1 successive Tonelli-Shanks square roots
// with z = the smallest quadratic non-residue, returning whatever root the loop at
// src/roots.rs:37-65 lands on) and the table layout (src/prime64.rs:183-218).  Everything else is
// exact integer arithmetic.
#pragma once
#include <cstdint>
#include <vector>

namespace cntt {
namespace host {

using u128 = unsigned __int128;

inline uint64_t mulmod(uint64_t a, uint64_t b, uint64_t m) { return (uint64_t)(((u128)a * b) % m); }

inline uint64_t powmod(uint64_t base, uint64_t exp, uint64_t m) {
    uint64_t acc = 1 % m;
    base %= m;
    for (; exp; exp >>= 1) {
        if (exp & 1) acc = mulmod(acc, base, m);
        base = mulmod(base, base, m);
    }
    return acc;
}

// deterministic Miller-Rabin for n < 2^64 with the first twelve primes as bases
// (same decision as src/prime.rs:76-126)
inline bool is_prime(uint64_t n) {
    static const uint64_t B[12] = {2, 3, 5, 7, 11, 13, 17, 19, 23, 29, 31, 37};
    if (n < 2) return false;
    for (uint64_t q : B)
        if (n % q == 0) return n == q;
    uint64_t d = n - 1;
    int s = 0;
    while ((d & 1) == 0) {
        d >>= 1;
        ++s;
    }
    for (uint64_t a : B) {
        uint64_t x = powmod(a, d, n);
        if (x == 1 || x == n - 1) continue;
        bool witness = true;
        for (int r = 1; r < s; ++r) {
            x = mulmod(x, x, n);
            if (x == n - 1) {
                witness = false;
                break;
            }
        }
        if (witness) return false;
    }
    return true;
}

// smallest n >= 2 with n^((p-1)/2) == p-1  (src/roots.rs:17-28)
inline bool smallest_non_residue(uint64_t p, uint64_t *z) {
    for (uint64_t n = 2; n < p; ++n)
        if (powmod(n, (p - 1) / 2, p) == p - 1) {
            *z = n;
            return true;
        }
    return false;
}

// Tonelli-Shanks exactly as src/roots.rs:31-66 (p - 1 = q * 2^s, z a non-residue)
inline bool tonelli_shanks(uint64_t p, uint64_t q, uint64_t s, uint64_t z, uint64_t n, uint64_t *out) {
    uint64_t m = s;
    uint64_t c = powmod(z, q, p);
    uint64_t t = powmod(n, q, p);
    uint64_t r = powmod(n, (q + 1) / 2, p);
    for (;;) {
        if (t == 0) {
            *out = 0;
            return true;
        }
        if (t == 1) {
            *out = r;
            return true;
        }
        uint64_t i = 0, t_pow = t;
        while (i < m) {
            t_pow = mulmod(t_pow, t_pow, p);
            ++i;
            if (t_pow == 1) break;
        }
        if (i == m) return false;
        const uint64_t b = powmod(c, (uint64_t)1 << (m - i - 1), p);
        m = i;
        c = mulmod(b, b, p);
        t = mulmod(t, c, p);
        r = mulmod(r, b, p);
    }
}

// src/roots.rs:68-91
inline bool primitive_root(uint64_t p, uint64_t degree, uint64_t *out) {
    int logd = 0;
    while (((uint64_t)1 << logd) < degree) ++logd;
    uint64_t q = p - 1, s = 0;
    while ((q & 1) == 0) {
        q >>= 1;
        ++s;
    }
    uint64_t z = 0;
    if (!smallest_non_residue(p, &z)) return false;
    uint64_t root = p - 1;
    for (int i = 0; i + 1 < logd; ++i) {
        uint64_t r = 0;
        if (!tonelli_shanks(p, q, s, z, root, &r)) return false;
        root = r;
    }
    *out = root;
    return true;
}

inline uint32_t bit_reverse(uint32_t nbits, uint32_t i) {  // src/lib.rs:118-121
    uint32_t r = 0;
    for (uint32_t b = 0; b < nbits; ++b) r |= ((i >> b) & 1u) << (nbits - 1 - b);
    return r;
}

// -x^-1 mod 2^64 for odd x (Newton)
inline uint64_t neg_inv_pow2(uint64_t x) {
    uint64_t inv = x;  // correct to 3 bits
    for (int i = 0; i < 6; ++i) inv *= 2 - x * inv;
    return (uint64_t)0 - inv;
}

// CLS_FP table entries: the centred representative of x mod p as an exact double (p < 2^50), and a double's bits
inline double centred(uint64_t x, uint64_t p) { return x > p / 2 ? -(double)(p - x) : (double)x; }
inline uint64_t double_bits(double d) {
    uint64_t u;
    static_assert(sizeof u == sizeof d, "IEEE double");
    __builtin_memcpy(&u, &d, sizeof u);
    return u;
}

}  // namespace host
}  // namespace cntt
