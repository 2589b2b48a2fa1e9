/*
 * cntt_oracle.c -- CPU ORACLE (TEST INFRASTRUCTURE ONLY, NOT THE PRODUCT).
 * See cntt_oracle.h for the parity pin.  Every function follows the cited
 * lines of the scalar (non-SIMD) path of the reference; the reference's own
 * tests assert that its AVX2/AVX-512 paths produce identical values
 * (e.g. src/native64.rs:1245-1293, src/prime64.rs:1564-1877).
 */
#define _GNU_SOURCE
#include "cntt_oracle.h"

#include <pthread.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

typedef orc_u128 u128;

/* ------------------------------------------------------------------------- */
/* src/lib.rs:118-121  bit_rev: i.reverse_bits() >> (usize::BITS - nbits)      */
size_t orc_bit_rev(uint32_t nbits, size_t i) {
    size_t r = 0;
    for (uint32_t b = 0; b < nbits; ++b) r |= ((i >> b) & 1u) << (nbits - 1 - b);
    return r;
}

/* src/fastdiv.rs:157-195 asserts Div32/Div64 equal plain `/` and `%`, so the
 * native operators are a faithful restatement of the Lemire division. */
static inline uint32_t mul_mod32(uint32_t p, uint32_t x, uint32_t y) { /* src/prime.rs:4-6 */
    return (uint32_t)(((uint64_t)x * y) % p);
}
uint64_t orc_mul_mod64(uint64_t p, uint64_t x, uint64_t y) { /* src/prime.rs:8-10 */
    return (uint64_t)(((u128)x * y) % p);
}

uint32_t orc_exp_mod32(uint32_t p, uint32_t base, uint32_t pow) { /* src/prime.rs:12-29 */
    if (pow == 0) return 1;
    uint32_t y = 1, x = base;
    while (pow > 1) {
        if (pow % 2 == 1) y = mul_mod32(p, x, y);
        x = mul_mod32(p, x, x);
        pow /= 2;
    }
    return mul_mod32(p, x, y);
}

uint64_t orc_exp_mod64(uint64_t p, uint64_t base, uint64_t pow) { /* src/prime.rs:31-48 */
    if (pow == 0) return 1;
    uint64_t y = 1, x = base;
    while (pow > 1) {
        if (pow % 2 == 1) y = orc_mul_mod64(p, x, y);
        x = orc_mul_mod64(p, x, x);
        pow /= 2;
    }
    return orc_mul_mod64(p, x, y);
}

static int miller_rabin_iter(uint64_t n, uint64_t s, uint64_t d, uint64_t a) { /* src/prime.rs:50-66 */
    uint64_t x = orc_exp_mod64(n, a, d);
    uint64_t n_minus_1 = n - 1;
    if (x == 1 || x == n_minus_1) return 1;
    uint64_t count = 0;
    while (count < s - 1) {
        x = orc_mul_mod64(n, x, x);
        if (x == n_minus_1) return 1;
        count += 1;
    }
    return 0;
}

int orc_is_prime64(uint64_t n) { /* src/prime.rs:76-126 */
    if (n < 2) return 0;
    static const uint64_t small[12] = {2, 3, 5, 7, 11, 13, 17, 19, 23, 29, 31, 37};
    for (int i = 0; i < 12; ++i)
        if (n % small[i] == 0) return n == small[i];
    uint64_t s = 0, d = n - 1;
    while (d % 2 == 0) {
        s += 1;
        d /= 2;
    }
    for (int i = 0; i < 12; ++i)
        if (!miller_rabin_iter(n, s, d, small[i])) return 0;
    return 1;
}

int orc_largest_prime_in_arithmetic_progression64(uint64_t factor, uint64_t offset, uint64_t lo,
                                                  uint64_t hi, uint64_t *out) { /* src/prime.rs:130-180 */
    if (lo > hi) return 0;
    uint64_t a = factor, b = offset;
    if (b > hi) return 0;
    if (a == 0) {
        if (lo <= b && b <= hi && orc_is_prime64(b)) {
            *out = b;
            return 1;
        }
        return 0;
    }
    uint64_t m = lo > b ? lo : b;
    uint64_t x_lo = (m - b) / a;
    if ((m - b) % a != 0) x_lo += 1;
    uint64_t x_hi = (hi - b) / a;
    uint64_t x = x_hi;
    int in_range = 1;
    while (in_range) {
        uint64_t val = a * x + b;
        if (orc_is_prime64(val)) {
            *out = val;
            return 1;
        }
        if (x == x_lo)
            in_range = 0;
        else
            x -= 1;
    }
    return 0;
}

/* src/roots.rs:6-15 */
static void get_q_s64(uint64_t p, uint64_t *q, uint64_t *s) {
    uint64_t qq = p - 1, ss = 0;
    while (qq % 2 == 0) {
        qq /= 2;
        ss += 1;
    }
    *q = qq;
    *s = ss;
}

int orc_get_z64(uint64_t p, uint64_t *z) { /* src/roots.rs:17-28 */
    uint64_t n = 2;
    while (n < p) {
        if (orc_exp_mod64(p, n, (p - 1) / 2) == p - 1) {
            *z = n;
            return 1;
        }
        n += 1;
    }
    return 0;
}

int orc_sqrt_mod_ex64(uint64_t p, uint64_t q, uint64_t s, uint64_t z, uint64_t n, uint64_t *out) {
    /* src/roots.rs:31-66 -- followed literally, no normalisation of the returned root */
    uint64_t m = s;
    uint64_t c = orc_exp_mod64(p, z, q);
    uint64_t t = orc_exp_mod64(p, n, q);
    uint64_t r = orc_exp_mod64(p, n, (q + 1) / 2);
    for (;;) {
        if (t == 0) {
            *out = 0;
            return 1;
        }
        if (t == 1) {
            *out = r;
            return 1;
        }
        uint64_t i = 0;
        uint64_t t_pow = t;
        while (i < m) {
            t_pow = orc_mul_mod64(p, t_pow, t_pow);
            i += 1;
            if (t_pow == 1) break;
        }
        if (i == m) return 0; /* None */
        uint64_t b = orc_exp_mod64(p, c, (uint64_t)1 << (m - i - 1));
        m = i;
        c = orc_mul_mod64(p, b, b);
        t = orc_mul_mod64(p, t, c);
        r = orc_mul_mod64(p, r, b);
    }
}

int orc_find_primitive_root64(uint64_t p, uint64_t degree, uint64_t *root_out) { /* src/roots.rs:68-91 */
    /* assert!(degree.is_power_of_two()); assert!(degree > 1); */
    uint32_t n = (uint32_t)__builtin_ctzll(degree);
    uint64_t root = p - 1;
    uint64_t q, s, z;
    get_q_s64(p, &q, &s);
    if (!orc_get_z64(p, &z)) return 0;
    for (uint32_t i = 0; i + 1 < n; ++i) {
        uint64_t r;
        if (!orc_sqrt_mod_ex64(p, q, s, z, root, &r)) return 0;
        root = r;
    }
    *root_out = root;
    return 1;
}

/* ========================================================================= */
/* prime64                                                                   */
/* ========================================================================= */
#define RECURSION_THRESHOLD64 1024 /* src/prime64.rs:7 */
#define RECURSION_THRESHOLD32 2048 /* src/prime32.rs:12 */
#define SOLINAS_P 0xFFFFFFFF00000001ull /* src/prime64/generic_solinas.rs:38-40 */

static inline uint64_t min64(uint64_t a, uint64_t b) { return a < b ? a : b; }
static inline uint32_t min32(uint32_t a, uint32_t b) { return a < b ? a : b; }

static void *xaligned(size_t bytes) {
    void *p = NULL;
    if (bytes == 0) bytes = 64;
    if (posix_memalign(&p, 64, (bytes + 63) & ~(size_t)63) != 0) return NULL;
    memset(p, 0, bytes);
    return p;
}

/* src/prime64.rs:158-181 (bits < 0: no shoup tables) and :183-218 */
static void init_negacyclic_twiddles64(uint64_t p, size_t n, int shoup, uint64_t *twid,
                                       uint64_t *twid_shoup, uint64_t *inv_twid,
                                       uint64_t *inv_twid_shoup) {
    uint64_t w = 0;
    orc_find_primitive_root64(p, 2 * (uint64_t)n, &w);
    size_t k = 0;
    uint64_t wk = 1;
    uint32_t nbits = (uint32_t)__builtin_ctzll(n);
    while (k < n) {
        size_t fwd_idx = orc_bit_rev(nbits, k);
        uint64_t wk_shoup = 0;
        if (shoup) wk_shoup = (uint64_t)((((u128)wk) << 64) / p);
        twid[fwd_idx] = wk;
        if (shoup) twid_shoup[fwd_idx] = wk_shoup;
        size_t inv_idx = orc_bit_rev(nbits, (n - k) % n);
        if (k == 0) {
            inv_twid[inv_idx] = wk;
            if (shoup) inv_twid_shoup[inv_idx] = wk_shoup;
        } else {
            uint64_t x = p - wk;
            inv_twid[inv_idx] = x;
            if (shoup) inv_twid_shoup[inv_idx] = (uint64_t)((((u128)x) << 64) / p);
        }
        wk = (uint64_t)(((u128)wk * w) % p);
        k += 1;
    }
}

static uint32_t ilog2_64(uint64_t x) { return 63u - (uint32_t)__builtin_clzll(x); }

orc_plan64 *orc_plan64_try_new(size_t n, uint64_t p, int *panicked) { /* src/prime64.rs:704-771 */
    if (panicked) *panicked = 0;
    if (p <= 1) { /* Div64::new asserts divisor > 1: src/fastdiv.rs:99 */
        if (panicked) *panicked = 1;
        return NULL;
    }
    uint64_t root;
    if (n < 16 || (n & (n - 1)) != 0 || !orc_is_prime64(p) ||
        !orc_find_primitive_root64(p, 2 * (uint64_t)n, &root))
        return NULL;
    /* has_ifma == false: the 52-bit Shoup shift is an internal detail of the IFMA CPU path;
     * the scalar/AVX2/AVX-512 paths all use bits = 64 (src/prime64.rs:716-725). */
    const uint32_t bits = 64;
    orc_plan64 *plan = (orc_plan64 *)calloc(1, sizeof(*plan));
    plan->n = n;
    plan->p = p;
    plan->twid = (uint64_t *)xaligned(n * 8);
    plan->inv_twid = (uint64_t *)xaligned(n * 8);
    int shoup = p < ((uint64_t)1 << 63);
    if (shoup) {
        plan->twid_shoup = (uint64_t *)xaligned(n * 8);
        plan->inv_twid_shoup = (uint64_t *)xaligned(n * 8);
    }
    init_negacyclic_twiddles64(p, n, shoup, plan->twid, plan->twid_shoup, plan->inv_twid,
                               plan->inv_twid_shoup);
    plan->n_inv_mod_p = orc_exp_mod64(p, (uint64_t)n, p - 2);
    plan->n_inv_mod_p_shoup = (uint64_t)((((u128)plan->n_inv_mod_p) << bits) / p);
    plan->big_q = ilog2_64(p) + 1;
    uint64_t big_l = plan->big_q + (bits - 1);
    /* (1u128 << big_l) / p as u64; for p >= 2^63 big_l = 127 and the value is unused */
    plan->p_barrett = (uint64_t)((((u128)1) << big_l) / p);
    return plan;
}

void orc_plan64_free(orc_plan64 *plan) {
    if (!plan) return;
    free(plan->twid);
    free(plan->twid_shoup);
    free(plan->inv_twid);
    free(plan->inv_twid_shoup);
    free(plan);
}

/* ---- butterflies.  class 62: src/prime64/less_than_62bit.rs:117-154, :271-310
 *                    class 63: src/prime64/less_than_63bit.rs:117-154, :214-232 ---- */
typedef struct {
    uint64_t a, b;
} pair64;

static inline pair64 fwd_bfly62(uint64_t z0, uint64_t z1, uint64_t w, uint64_t ws, uint64_t p,
                                uint64_t neg_p, uint64_t two_p) {
    (void)p;
    z0 = min64(z0, z0 - two_p);
    uint64_t shoup_q = (uint64_t)(((u128)z1 * ws) >> 64);
    uint64_t t = z1 * w + shoup_q * neg_p;
    return (pair64){z0 + t, z0 - t + two_p};
}
static inline pair64 fwd_last_bfly62(uint64_t z0, uint64_t z1, uint64_t w, uint64_t ws, uint64_t p,
                                     uint64_t neg_p, uint64_t two_p) {
    z0 = min64(z0, z0 - two_p);
    z0 = min64(z0, z0 - p);
    uint64_t shoup_q = (uint64_t)(((u128)z1 * ws) >> 64);
    uint64_t t = z1 * w + shoup_q * neg_p;
    t = min64(t, t - p);
    uint64_t r0 = z0 + t, r1 = z0 - t + p;
    return (pair64){min64(r0, r0 - p), min64(r1, r1 - p)};
}
static inline pair64 inv_bfly62(uint64_t z0, uint64_t z1, uint64_t w, uint64_t ws, uint64_t p,
                                uint64_t neg_p, uint64_t two_p) {
    (void)p;
    uint64_t y0 = z0 + z1;
    y0 = min64(y0, y0 - two_p);
    uint64_t t = z0 - z1 + two_p;
    uint64_t shoup_q = (uint64_t)(((u128)t * ws) >> 64);
    uint64_t y1 = t * w + shoup_q * neg_p;
    return (pair64){y0, y1};
}
static inline pair64 inv_last_bfly62(uint64_t z0, uint64_t z1, uint64_t w, uint64_t ws, uint64_t p,
                                     uint64_t neg_p, uint64_t two_p) {
    uint64_t y0 = z0 + z1;
    y0 = min64(y0, y0 - two_p);
    y0 = min64(y0, y0 - p);
    uint64_t t = z0 - z1 + two_p;
    uint64_t shoup_q = (uint64_t)(((u128)t * ws) >> 64);
    uint64_t y1 = t * w + shoup_q * neg_p;
    y1 = min64(y1, y1 - p);
    return (pair64){y0, y1};
}
static inline pair64 fwd_bfly63(uint64_t z0, uint64_t z1, uint64_t w, uint64_t ws, uint64_t p,
                                uint64_t neg_p, uint64_t two_p) {
    (void)two_p;
    z0 = min64(z0, z0 - p);
    uint64_t shoup_q = (uint64_t)(((u128)z1 * ws) >> 64);
    uint64_t t = z1 * w + shoup_q * neg_p;
    t = min64(t, t - p);
    return (pair64){z0 + t, z0 - t + p};
}
static inline pair64 fwd_last_bfly63(uint64_t z0, uint64_t z1, uint64_t w, uint64_t ws, uint64_t p,
                                     uint64_t neg_p, uint64_t two_p) {
    (void)two_p;
    z0 = min64(z0, z0 - p);
    uint64_t shoup_q = (uint64_t)(((u128)z1 * ws) >> 64);
    uint64_t t = z1 * w + shoup_q * neg_p;
    t = min64(t, t - p);
    uint64_t r0 = z0 + t, r1 = z0 - t + p;
    return (pair64){min64(r0, r0 - p), min64(r1, r1 - p)};
}
static inline pair64 inv_bfly63(uint64_t z0, uint64_t z1, uint64_t w, uint64_t ws, uint64_t p,
                                uint64_t neg_p, uint64_t two_p) {
    (void)two_p;
    uint64_t y0 = z0 + z1;
    y0 = min64(y0, y0 - p);
    uint64_t t = z0 - z1 + p;
    uint64_t shoup_q = (uint64_t)(((u128)t * ws) >> 64);
    uint64_t y1 = t * w + shoup_q * neg_p;
    y1 = min64(y1, y1 - p);
    return (pair64){y0, y1};
}

typedef pair64 (*bfly64_fn)(uint64_t, uint64_t, uint64_t, uint64_t, uint64_t, uint64_t, uint64_t);

/* src/prime64/shoup.rs:544-615 */
static void fwd_breadth_first_scalar64(uint64_t p, uint64_t *data, size_t n, const uint64_t *twid,
                                       const uint64_t *twid_shoup, size_t depth, size_t half,
                                       bfly64_fn butterfly, bfly64_fn last_butterfly) {
    size_t t = n, m = 1;
    size_t w_idx = (m << depth) + half * m;
    uint64_t neg_p = (uint64_t)0 - p, two_p = 2 * p;
    while (m < n) {
        t /= 2;
        const uint64_t *w = twid + w_idx, *ws = twid_shoup + w_idx;
        bfly64_fn f = (t == 1) ? last_butterfly : butterfly;
        for (size_t blk = 0; blk < n / (2 * t); ++blk) {
            uint64_t *z0 = data + blk * 2 * t, *z1 = z0 + t;
            for (size_t j = 0; j < t; ++j) {
                pair64 r = f(z0[j], z1[j], w[blk], ws[blk], p, neg_p, two_p);
                z0[j] = r.a;
                z1[j] = r.b;
            }
        }
        m *= 2;
        w_idx *= 2;
    }
}

/* src/prime64/shoup.rs:617-706 */
static void fwd_depth_first_scalar64(uint64_t p, uint64_t *data, size_t n, const uint64_t *twid,
                                     const uint64_t *twid_shoup, size_t depth, size_t half,
                                     bfly64_fn butterfly, bfly64_fn last_butterfly) {
    if (n <= RECURSION_THRESHOLD64) {
        fwd_breadth_first_scalar64(p, data, n, twid, twid_shoup, depth, half, butterfly, last_butterfly);
        return;
    }
    size_t t = n / 2, m = 1;
    size_t w_idx = (m << depth) + m * half;
    uint64_t neg_p = (uint64_t)0 - p, two_p = 2 * p;
    uint64_t w = twid[w_idx], ws = twid_shoup[w_idx];
    for (size_t j = 0; j < t; ++j) {
        pair64 r = butterfly(data[j], data[t + j], w, ws, p, neg_p, two_p);
        data[j] = r.a;
        data[t + j] = r.b;
    }
    fwd_depth_first_scalar64(p, data, n / 2, twid, twid_shoup, depth + 1, half * 2, butterfly, last_butterfly);
    fwd_depth_first_scalar64(p, data + n / 2, n / 2, twid, twid_shoup, depth + 1, half * 2 + 1, butterfly,
                             last_butterfly);
}

/* src/prime64/shoup.rs:1306-1377 */
static void inv_breadth_first_scalar64(uint64_t p, uint64_t *data, size_t n, const uint64_t *inv_twid,
                                       const uint64_t *inv_twid_shoup, size_t depth, size_t half,
                                       bfly64_fn butterfly, bfly64_fn last_butterfly) {
    size_t t = 1, m = n;
    size_t w_idx = (m << depth) + half * m;
    uint64_t neg_p = (uint64_t)0 - p, two_p = 2 * p;
    while (m > 1) {
        m /= 2;
        w_idx /= 2;
        const uint64_t *w = inv_twid + w_idx, *ws = inv_twid_shoup + w_idx;
        bfly64_fn f = (m == 1) ? last_butterfly : butterfly;
        for (size_t blk = 0; blk < n / (2 * t); ++blk) {
            uint64_t *z0 = data + blk * 2 * t, *z1 = z0 + t;
            for (size_t j = 0; j < t; ++j) {
                pair64 r = f(z0[j], z1[j], w[blk], ws[blk], p, neg_p, two_p);
                z0[j] = r.a;
                z1[j] = r.b;
            }
        }
        t *= 2;
    }
}

/* src/prime64/shoup.rs:1379-1468 */
static void inv_depth_first_scalar64(uint64_t p, uint64_t *data, size_t n, const uint64_t *inv_twid,
                                     const uint64_t *inv_twid_shoup, size_t depth, size_t half,
                                     bfly64_fn butterfly, bfly64_fn last_butterfly) {
    if (n <= RECURSION_THRESHOLD64) {
        inv_breadth_first_scalar64(p, data, n, inv_twid, inv_twid_shoup, depth, half, butterfly, last_butterfly);
        return;
    }
    inv_depth_first_scalar64(p, data, n / 2, inv_twid, inv_twid_shoup, depth + 1, half * 2, butterfly, butterfly);
    inv_depth_first_scalar64(p, data + n / 2, n / 2, inv_twid, inv_twid_shoup, depth + 1, half * 2 + 1,
                             butterfly, butterfly);
    size_t t = n / 2, m = 1;
    size_t w_idx = (m << depth) + m * half;
    uint64_t neg_p = (uint64_t)0 - p, two_p = 2 * p;
    uint64_t w = inv_twid[w_idx], ws = inv_twid_shoup[w_idx];
    for (size_t j = 0; j < t; ++j) {
        pair64 r = last_butterfly(data[j], data[t + j], w, ws, p, neg_p, two_p);
        data[j] = r.a;
        data[t + j] = r.b;
    }
}

/* ---- generic / Solinas (src/prime64/generic_solinas.rs) ---- */
static inline uint64_t gen_add64(uint64_t p, uint64_t a, uint64_t b) { /* :46-58 / :80-89 */
    uint64_t neg_b = p - b;
    return (a >= neg_b) ? a - neg_b : a + b;
}
static inline uint64_t gen_sub64(uint64_t p, uint64_t a, uint64_t b) { /* :60-69 / :91-100 */
    uint64_t neg_b = p - b;
    return (a >= b) ? a - b : a + neg_b;
}
static inline uint64_t solinas_mul(uint64_t a, uint64_t b) { /* :102-128 */
    const uint64_t p = SOLINAS_P;
    u128 wide = (u128)a * b;
    uint64_t lo = (uint64_t)wide;
    uint64_t hi = (uint64_t)(wide >> 64);
    uint64_t mid = hi & 0x00000000FFFFFFFFull;
    hi = (hi & 0xFFFFFFFF00000000ull) >> 32;
    uint64_t low2 = lo - hi;
    if (hi > lo) low2 += p;
    uint64_t product = mid << 32;
    product -= mid;
    uint64_t result = low2 + product;
    if ((result < product) || (result >= p)) result -= p;
    return result;
}
static inline uint64_t gen_mul64(uint64_t p, int solinas, uint64_t a, uint64_t b) {
    return solinas ? solinas_mul(a, b) : (uint64_t)(((u128)a * b) % p); /* :71-75 */
}

/* :449-481 */
static void gen_fwd_breadth_first64(uint64_t *data, size_t n, uint64_t p, int solinas, const uint64_t *twid,
                                    size_t depth, size_t half) {
    size_t t = n / 2, m = 1;
    size_t w_idx = (m << depth) + half * m;
    while (m < n) {
        const uint64_t *w = twid + w_idx;
        for (size_t blk = 0; blk < n / (2 * t); ++blk) {
            uint64_t *z0 = data + blk * 2 * t, *z1 = z0 + t;
            uint64_t w1 = w[blk];
            for (size_t j = 0; j < t; ++j) {
                uint64_t z1w = gen_mul64(p, solinas, z1[j], w1);
                uint64_t a = gen_add64(p, z0[j], z1w), b = gen_sub64(p, z0[j], z1w);
                z0[j] = a;
                z1[j] = b;
            }
        }
        t /= 2;
        m *= 2;
        w_idx *= 2;
    }
}
/* :1338-1386 */
static void gen_fwd_depth_first64(uint64_t *data, size_t n, uint64_t p, int solinas, const uint64_t *twid,
                                  size_t depth, size_t half) {
    if (n <= RECURSION_THRESHOLD64) {
        gen_fwd_breadth_first64(data, n, p, solinas, twid, depth, half);
        return;
    }
    size_t t = n / 2, m = 1;
    size_t w_idx = (m << depth) + m * half;
    uint64_t w1 = twid[w_idx];
    for (size_t j = 0; j < t; ++j) {
        uint64_t z1w = gen_mul64(p, solinas, data[t + j], w1);
        uint64_t a = gen_add64(p, data[j], z1w), b = gen_sub64(p, data[j], z1w);
        data[j] = a;
        data[t + j] = b;
    }
    gen_fwd_depth_first64(data, n / 2, p, solinas, twid, depth + 1, half * 2);
    gen_fwd_depth_first64(data + n / 2, n / 2, p, solinas, twid, depth + 1, half * 2 + 1);
}
/* :483-513 */
static void gen_inv_breadth_first64(uint64_t *data, size_t n, uint64_t p, int solinas,
                                    const uint64_t *inv_twid, size_t depth, size_t half) {
    size_t t = 1, m = n;
    size_t w_idx = (m << depth) + half * m;
    while (m > 1) {
        m /= 2;
        w_idx /= 2;
        const uint64_t *w = inv_twid + w_idx;
        for (size_t blk = 0; blk < n / (2 * t); ++blk) {
            uint64_t *z0 = data + blk * 2 * t, *z1 = z0 + t;
            uint64_t w1 = w[blk];
            for (size_t j = 0; j < t; ++j) {
                uint64_t a = gen_add64(p, z0[j], z1[j]);
                uint64_t b = gen_mul64(p, solinas, gen_sub64(p, z0[j], z1[j]), w1);
                z0[j] = a;
                z1[j] = b;
            }
        }
        t *= 2;
    }
}
/* :515-561 */
static void gen_inv_depth_first64(uint64_t *data, size_t n, uint64_t p, int solinas, const uint64_t *inv_twid,
                                  size_t depth, size_t half) {
    if (n <= RECURSION_THRESHOLD64) {
        gen_inv_breadth_first64(data, n, p, solinas, inv_twid, depth, half);
        return;
    }
    gen_inv_depth_first64(data, n / 2, p, solinas, inv_twid, depth + 1, half * 2);
    gen_inv_depth_first64(data + n / 2, n / 2, p, solinas, inv_twid, depth + 1, half * 2 + 1);
    size_t t = n / 2, m = 1;
    size_t w_idx = (m << depth) + m * half;
    uint64_t w1 = inv_twid[w_idx];
    for (size_t j = 0; j < t; ++j) {
        uint64_t a = gen_add64(p, data[j], data[t + j]);
        uint64_t b = gen_mul64(p, solinas, gen_sub64(p, data[j], data[t + j]), w1);
        data[j] = a;
        data[t + j] = b;
    }
}

void orc_plan64_fwd(const orc_plan64 *plan, uint64_t *buf) { /* src/prime64.rs:794-865 */
    uint64_t p = plan->p;
    size_t n = plan->n;
    if (p < ((uint64_t)1 << 62))
        fwd_depth_first_scalar64(p, buf, n, plan->twid, plan->twid_shoup, 0, 0, fwd_bfly62, fwd_last_bfly62);
    else if (p < ((uint64_t)1 << 63))
        fwd_depth_first_scalar64(p, buf, n, plan->twid, plan->twid_shoup, 0, 0, fwd_bfly63, fwd_last_bfly63);
    else
        gen_fwd_depth_first64(buf, n, p, p == SOLINAS_P, plan->twid, 0, 0);
}

void orc_plan64_inv(const orc_plan64 *plan, uint64_t *buf) { /* src/prime64.rs:872-943 */
    uint64_t p = plan->p;
    size_t n = plan->n;
    if (p < ((uint64_t)1 << 62))
        inv_depth_first_scalar64(p, buf, n, plan->inv_twid, plan->inv_twid_shoup, 0, 0, inv_bfly62,
                                 inv_last_bfly62);
    else if (p < ((uint64_t)1 << 63)) /* 63-bit class has no separate last butterfly: less_than_63bit.rs */
        inv_depth_first_scalar64(p, buf, n, plan->inv_twid, plan->inv_twid_shoup, 0, 0, inv_bfly63, inv_bfly63);
    else
        gen_inv_depth_first64(buf, n, p, p == SOLINAS_P, plan->inv_twid, 0, 0);
}

void orc_plan64_mul_assign_normalize(const orc_plan64 *plan, uint64_t *lhs, const uint64_t *rhs, size_t len) {
    uint64_t p = plan->p;
    if (p < ((uint64_t)1 << 63)) { /* src/prime64.rs:534-559 */
        uint64_t big_q_m1 = plan->big_q - 1;
        for (size_t i = 0; i < len; ++i) {
            u128 d = (u128)lhs[i] * rhs[i];
            uint64_t c1 = (uint64_t)(d >> big_q_m1);
            uint64_t c3 = (uint64_t)(((u128)c1 * plan->p_barrett) >> 64);
            uint64_t prod = (uint64_t)d - p * c3;
            uint64_t shoup_q = (uint64_t)(((u128)prod * plan->n_inv_mod_p_shoup) >> 64);
            uint64_t t = prod * plan->n_inv_mod_p - shoup_q * p;
            lhs[i] = min64(t, t - p);
        }
    } else { /* :1013-1032 */
        int solinas = p == SOLINAS_P;
        for (size_t i = 0; i < len; ++i) {
            uint64_t prod = gen_mul64(p, solinas, lhs[i], rhs[i]);
            lhs[i] = gen_mul64(p, solinas, prod, plan->n_inv_mod_p);
        }
    }
}

void orc_plan64_normalize(const orc_plan64 *plan, uint64_t *values, size_t len) {
    uint64_t p = plan->p;
    if (p < ((uint64_t)1 << 63)) { /* src/prime64.rs:690-699 */
        for (size_t i = 0; i < len; ++i) {
            uint64_t val = values[i];
            uint64_t shoup_q = (uint64_t)(((u128)val * plan->n_inv_mod_p_shoup) >> 64);
            uint64_t t = val * plan->n_inv_mod_p - shoup_q * p;
            values[i] = min64(t, t - p);
        }
    } else { /* :1068-1081 */
        int solinas = p == SOLINAS_P;
        for (size_t i = 0; i < len; ++i) values[i] = gen_mul64(p, solinas, values[i], plan->n_inv_mod_p);
    }
}

void orc_plan64_mul_accumulate(const orc_plan64 *plan, uint64_t *acc, const uint64_t *lhs, const uint64_t *rhs,
                               size_t len) {
    uint64_t p = plan->p;
    if (p < ((uint64_t)1 << 63)) { /* src/prime64.rs:561-584 */
        uint64_t big_q_m1 = plan->big_q - 1;
        for (size_t i = 0; i < len; ++i) {
            u128 d = (u128)lhs[i] * rhs[i];
            uint64_t c1 = (uint64_t)(d >> big_q_m1);
            uint64_t c3 = (uint64_t)(((u128)c1 * plan->p_barrett) >> 64);
            uint64_t prod = (uint64_t)d - p * c3;
            prod = min64(prod, prod - p);
            uint64_t acc_ = prod + acc[i];
            acc[i] = min64(acc_, acc_ - p);
        }
    } else { /* :1116-1127 */
        int solinas = p == SOLINAS_P;
        for (size_t i = 0; i < len; ++i) {
            uint64_t prod = gen_mul64(p, solinas, lhs[i], rhs[i]);
            acc[i] = gen_add64(p, acc[i], prod);
        }
    }
}

/* ========================================================================= */
/* prime32                                                                   */
/* ========================================================================= */
/* src/prime32.rs:223-282 */
static void init_negacyclic_twiddles32(uint32_t p, size_t n, int shoup, uint32_t *twid, uint32_t *twid_shoup,
                                       uint32_t *inv_twid, uint32_t *inv_twid_shoup) {
    uint64_t w64 = 0;
    orc_find_primitive_root64((uint64_t)p, 2 * (uint64_t)n, &w64);
    uint32_t w = (uint32_t)w64;
    size_t k = 0;
    uint32_t wk = 1;
    uint32_t nbits = (uint32_t)__builtin_ctzll(n);
    while (k < n) {
        size_t fwd_idx = orc_bit_rev(nbits, k);
        uint32_t wk_shoup = 0;
        if (shoup) wk_shoup = (uint32_t)((((uint64_t)wk) << 32) / p);
        twid[fwd_idx] = wk;
        if (shoup) twid_shoup[fwd_idx] = wk_shoup;
        size_t inv_idx = orc_bit_rev(nbits, (n - k) % n);
        if (k == 0) {
            inv_twid[inv_idx] = wk;
            if (shoup) inv_twid_shoup[inv_idx] = wk_shoup;
        } else {
            uint32_t x = p - wk;
            inv_twid[inv_idx] = x;
            if (shoup) inv_twid_shoup[inv_idx] = (uint32_t)((((uint64_t)x) << 32) / p);
        }
        wk = (uint32_t)(((uint64_t)wk * w) % p);
        k += 1;
    }
}

orc_plan32 *orc_plan32_try_new(size_t n, uint32_t p, int *panicked) { /* src/prime32.rs:630-686 */
    if (panicked) *panicked = 0;
    if (p <= 1) { /* Div32::new asserts divisor > 1: src/fastdiv.rs:48-49 */
        if (panicked) *panicked = 1;
        return NULL;
    }
    uint64_t root;
    if (n < 32 || (n & (n - 1)) != 0 || !orc_is_prime64((uint64_t)p) ||
        !orc_find_primitive_root64((uint64_t)p, 2 * (uint64_t)n, &root))
        return NULL;
    orc_plan32 *plan = (orc_plan32 *)calloc(1, sizeof(*plan));
    plan->n = n;
    plan->p = p;
    plan->twid = (uint32_t *)xaligned(n * 4);
    plan->inv_twid = (uint32_t *)xaligned(n * 4);
    int shoup = p < ((uint32_t)1 << 31);
    if (shoup) {
        plan->twid_shoup = (uint32_t *)xaligned(n * 4);
        plan->inv_twid_shoup = (uint32_t *)xaligned(n * 4);
    }
    init_negacyclic_twiddles32(p, n, shoup, plan->twid, plan->twid_shoup, plan->inv_twid, plan->inv_twid_shoup);
    plan->n_inv_mod_p = orc_exp_mod32(p, (uint32_t)n, p - 2);
    plan->n_inv_mod_p_shoup = (uint32_t)((((uint64_t)plan->n_inv_mod_p) << 32) / p);
    plan->big_q = (31u - (uint32_t)__builtin_clz(p)) + 1;
    uint32_t big_l = plan->big_q + 31;
    plan->p_barrett = (uint32_t)((((uint64_t)1) << big_l) / p);
    return plan;
}

void orc_plan32_free(orc_plan32 *plan) {
    if (!plan) return;
    free(plan->twid);
    free(plan->twid_shoup);
    free(plan->inv_twid);
    free(plan->inv_twid_shoup);
    free(plan);
}

typedef struct {
    uint32_t a, b;
} pair32;
typedef pair32 (*bfly32_fn)(uint32_t, uint32_t, uint32_t, uint32_t, uint32_t, uint32_t, uint32_t);

/* src/prime32/less_than_30bit.rs:115-153, :265-303 */
static inline pair32 fwd_bfly30(uint32_t z0, uint32_t z1, uint32_t w, uint32_t ws, uint32_t p, uint32_t neg_p,
                                uint32_t two_p) {
    (void)p;
    z0 = min32(z0, z0 - two_p);
    uint32_t shoup_q = (uint32_t)(((uint64_t)z1 * ws) >> 32);
    uint32_t t = z1 * w + shoup_q * neg_p;
    return (pair32){z0 + t, z0 - t + two_p};
}
static inline pair32 fwd_last_bfly30(uint32_t z0, uint32_t z1, uint32_t w, uint32_t ws, uint32_t p,
                                     uint32_t neg_p, uint32_t two_p) {
    z0 = min32(z0, z0 - two_p);
    z0 = min32(z0, z0 - p);
    uint32_t shoup_q = (uint32_t)(((uint64_t)z1 * ws) >> 32);
    uint32_t t = z1 * w + shoup_q * neg_p;
    t = min32(t, t - p);
    uint32_t r0 = z0 + t, r1 = z0 - t + p;
    return (pair32){min32(r0, r0 - p), min32(r1, r1 - p)};
}
static inline pair32 inv_bfly30(uint32_t z0, uint32_t z1, uint32_t w, uint32_t ws, uint32_t p, uint32_t neg_p,
                                uint32_t two_p) {
    (void)p;
    uint32_t y0 = z0 + z1;
    y0 = min32(y0, y0 - two_p);
    uint32_t t = z0 - z1 + two_p;
    uint32_t shoup_q = (uint32_t)(((uint64_t)t * ws) >> 32);
    uint32_t y1 = t * w + shoup_q * neg_p;
    return (pair32){y0, y1};
}
static inline pair32 inv_last_bfly30(uint32_t z0, uint32_t z1, uint32_t w, uint32_t ws, uint32_t p,
                                     uint32_t neg_p, uint32_t two_p) {
    uint32_t y0 = z0 + z1;
    y0 = min32(y0, y0 - two_p);
    uint32_t t = z0 - z1 + two_p;
    uint32_t shoup_q = (uint32_t)(((uint64_t)t * ws) >> 32);
    uint32_t y1 = t * w + shoup_q * neg_p;
    return (pair32){min32(y0, y0 - p), min32(y1, y1 - p)};
}
/* src/prime32/less_than_31bit.rs:117-156, :214-233 */
static inline pair32 fwd_bfly31(uint32_t z0, uint32_t z1, uint32_t w, uint32_t ws, uint32_t p, uint32_t neg_p,
                                uint32_t two_p) {
    (void)two_p;
    z0 = min32(z0, z0 - p);
    uint32_t shoup_q = (uint32_t)(((uint64_t)z1 * ws) >> 32);
    uint32_t t = z1 * w + shoup_q * neg_p;
    t = min32(t, t - p);
    return (pair32){z0 + t, z0 - t + p};
}
static inline pair32 fwd_last_bfly31(uint32_t z0, uint32_t z1, uint32_t w, uint32_t ws, uint32_t p,
                                     uint32_t neg_p, uint32_t two_p) {
    (void)two_p;
    z0 = min32(z0, z0 - p);
    uint32_t shoup_q = (uint32_t)(((uint64_t)z1 * ws) >> 32);
    uint32_t t = z1 * w + shoup_q * neg_p;
    t = min32(t, t - p);
    uint32_t r0 = z0 + t, r1 = z0 - t + p;
    return (pair32){min32(r0, r0 - p), min32(r1, r1 - p)};
}
static inline pair32 inv_bfly31(uint32_t z0, uint32_t z1, uint32_t w, uint32_t ws, uint32_t p, uint32_t neg_p,
                                uint32_t two_p) {
    (void)two_p;
    uint32_t y0 = z0 + z1;
    y0 = min32(y0, y0 - p);
    uint32_t t = z0 - z1 + p;
    uint32_t shoup_q = (uint32_t)(((uint64_t)t * ws) >> 32);
    uint32_t y1 = t * w + shoup_q * neg_p;
    y1 = min32(y1, y1 - p);
    return (pair32){y0, y1};
}

/* src/prime32/shoup.rs:582-635 */
static void fwd_breadth_first_scalar32(uint32_t p, uint32_t *data, size_t n, const uint32_t *twid,
                                       const uint32_t *twid_shoup, size_t depth, size_t half,
                                       bfly32_fn butterfly, bfly32_fn last_butterfly) {
    size_t t = n, m = 1;
    size_t w_idx = (m << depth) + half * m;
    uint32_t neg_p = (uint32_t)0 - p, two_p = 2 * p;
    while (m < n) {
        t /= 2;
        const uint32_t *w = twid + w_idx, *ws = twid_shoup + w_idx;
        bfly32_fn f = (t == 1) ? last_butterfly : butterfly;
        for (size_t blk = 0; blk < n / (2 * t); ++blk) {
            uint32_t *z0 = data + blk * 2 * t, *z1 = z0 + t;
            for (size_t j = 0; j < t; ++j) {
                pair32 r = f(z0[j], z1[j], w[blk], ws[blk], p, neg_p, two_p);
                z0[j] = r.a;
                z1[j] = r.b;
            }
        }
        m *= 2;
        w_idx *= 2;
    }
}
/* src/prime32/shoup.rs:637-708 */
static void fwd_depth_first_scalar32(uint32_t p, uint32_t *data, size_t n, const uint32_t *twid,
                                     const uint32_t *twid_shoup, size_t depth, size_t half,
                                     bfly32_fn butterfly, bfly32_fn last_butterfly) {
    if (n <= RECURSION_THRESHOLD32) {
        fwd_breadth_first_scalar32(p, data, n, twid, twid_shoup, depth, half, butterfly, last_butterfly);
        return;
    }
    size_t t = n / 2, m = 1;
    size_t w_idx = (m << depth) + m * half;
    uint32_t neg_p = (uint32_t)0 - p, two_p = 2 * p;
    uint32_t w = twid[w_idx], ws = twid_shoup[w_idx];
    for (size_t j = 0; j < t; ++j) {
        pair32 r = butterfly(data[j], data[t + j], w, ws, p, neg_p, two_p);
        data[j] = r.a;
        data[t + j] = r.b;
    }
    fwd_depth_first_scalar32(p, data, n / 2, twid, twid_shoup, depth + 1, half * 2, butterfly, last_butterfly);
    fwd_depth_first_scalar32(p, data + n / 2, n / 2, twid, twid_shoup, depth + 1, half * 2 + 1, butterfly,
                             last_butterfly);
}
/* src/prime32/shoup.rs:1355-1408 */
static void inv_breadth_first_scalar32(uint32_t p, uint32_t *data, size_t n, const uint32_t *inv_twid,
                                       const uint32_t *inv_twid_shoup, size_t depth, size_t half,
                                       bfly32_fn butterfly, bfly32_fn last_butterfly) {
    size_t t = 1, m = n;
    size_t w_idx = (m << depth) + half * m;
    uint32_t neg_p = (uint32_t)0 - p, two_p = 2 * p;
    while (m > 1) {
        m /= 2;
        w_idx /= 2;
        const uint32_t *w = inv_twid + w_idx, *ws = inv_twid_shoup + w_idx;
        bfly32_fn f = (m == 1) ? last_butterfly : butterfly;
        for (size_t blk = 0; blk < n / (2 * t); ++blk) {
            uint32_t *z0 = data + blk * 2 * t, *z1 = z0 + t;
            for (size_t j = 0; j < t; ++j) {
                pair32 r = f(z0[j], z1[j], w[blk], ws[blk], p, neg_p, two_p);
                z0[j] = r.a;
                z1[j] = r.b;
            }
        }
        t *= 2;
    }
}
/* src/prime32/shoup.rs:1410-1481 */
static void inv_depth_first_scalar32(uint32_t p, uint32_t *data, size_t n, const uint32_t *inv_twid,
                                     const uint32_t *inv_twid_shoup, size_t depth, size_t half,
                                     bfly32_fn butterfly, bfly32_fn last_butterfly) {
    if (n <= RECURSION_THRESHOLD32) {
        inv_breadth_first_scalar32(p, data, n, inv_twid, inv_twid_shoup, depth, half, butterfly, last_butterfly);
        return;
    }
    inv_depth_first_scalar32(p, data, n / 2, inv_twid, inv_twid_shoup, depth + 1, half * 2, butterfly, butterfly);
    inv_depth_first_scalar32(p, data + n / 2, n / 2, inv_twid, inv_twid_shoup, depth + 1, half * 2 + 1,
                             butterfly, butterfly);
    size_t t = n / 2, m = 1;
    size_t w_idx = (m << depth) + m * half;
    uint32_t neg_p = (uint32_t)0 - p, two_p = 2 * p;
    uint32_t w = inv_twid[w_idx], ws = inv_twid_shoup[w_idx];
    for (size_t j = 0; j < t; ++j) {
        pair32 r = last_butterfly(data[j], data[t + j], w, ws, p, neg_p, two_p);
        data[j] = r.a;
        data[t + j] = r.b;
    }
}

/* src/prime32/generic.rs:9-31 */
static inline uint32_t gen_add32(uint32_t p, uint32_t a, uint32_t b) {
    uint32_t neg_b = p - b;
    return (a >= neg_b) ? a - neg_b : a + b;
}
static inline uint32_t gen_sub32(uint32_t p, uint32_t a, uint32_t b) {
    uint32_t neg_b = p - b;
    return (a >= b) ? a - b : a + neg_b;
}
static inline uint32_t gen_mul32(uint32_t p, uint32_t a, uint32_t b) { return (uint32_t)(((uint64_t)a * b) % p); }

/* src/prime32/generic.rs:228-260 */
static void gen_fwd_breadth_first32(uint32_t *data, size_t n, uint32_t p, const uint32_t *twid, size_t depth,
                                    size_t half) {
    size_t t = n / 2, m = 1;
    size_t w_idx = (m << depth) + half * m;
    while (m < n) {
        const uint32_t *w = twid + w_idx;
        for (size_t blk = 0; blk < n / (2 * t); ++blk) {
            uint32_t *z0 = data + blk * 2 * t, *z1 = z0 + t;
            uint32_t w1 = w[blk];
            for (size_t j = 0; j < t; ++j) {
                uint32_t z1w = gen_mul32(p, z1[j], w1);
                uint32_t a = gen_add32(p, z0[j], z1w), b = gen_sub32(p, z0[j], z1w);
                z0[j] = a;
                z1[j] = b;
            }
        }
        t /= 2;
        m *= 2;
        w_idx *= 2;
    }
}
/* src/prime32/generic.rs:262-310 */
static void gen_fwd_depth_first32(uint32_t *data, size_t n, uint32_t p, const uint32_t *twid, size_t depth,
                                  size_t half) {
    if (n <= RECURSION_THRESHOLD32) {
        gen_fwd_breadth_first32(data, n, p, twid, depth, half);
        return;
    }
    size_t t = n / 2, m = 1;
    size_t w_idx = (m << depth) + m * half;
    uint32_t w1 = twid[w_idx];
    for (size_t j = 0; j < t; ++j) {
        uint32_t z1w = gen_mul32(p, data[t + j], w1);
        uint32_t a = gen_add32(p, data[j], z1w), b = gen_sub32(p, data[j], z1w);
        data[j] = a;
        data[t + j] = b;
    }
    gen_fwd_depth_first32(data, n / 2, p, twid, depth + 1, half * 2);
    gen_fwd_depth_first32(data + n / 2, n / 2, p, twid, depth + 1, half * 2 + 1);
}
/* src/prime32/generic.rs:312-343 */
static void gen_inv_breadth_first32(uint32_t *data, size_t n, uint32_t p, const uint32_t *inv_twid,
                                    size_t depth, size_t half) {
    size_t t = 1, m = n;
    size_t w_idx = (m << depth) + half * m;
    while (m > 1) {
        m /= 2;
        w_idx /= 2;
        const uint32_t *w = inv_twid + w_idx;
        for (size_t blk = 0; blk < n / (2 * t); ++blk) {
            uint32_t *z0 = data + blk * 2 * t, *z1 = z0 + t;
            uint32_t w1 = w[blk];
            for (size_t j = 0; j < t; ++j) {
                uint32_t a = gen_add32(p, z0[j], z1[j]);
                uint32_t b = gen_mul32(p, gen_sub32(p, z0[j], z1[j]), w1);
                z0[j] = a;
                z1[j] = b;
            }
        }
        t *= 2;
    }
}
/* src/prime32/generic.rs:345-390 */
static void gen_inv_depth_first32(uint32_t *data, size_t n, uint32_t p, const uint32_t *inv_twid, size_t depth,
                                  size_t half) {
    if (n <= RECURSION_THRESHOLD32) {
        gen_inv_breadth_first32(data, n, p, inv_twid, depth, half);
        return;
    }
    gen_inv_depth_first32(data, n / 2, p, inv_twid, depth + 1, half * 2);
    gen_inv_depth_first32(data + n / 2, n / 2, p, inv_twid, depth + 1, half * 2 + 1);
    size_t t = n / 2, m = 1;
    size_t w_idx = (m << depth) + m * half;
    uint32_t w1 = inv_twid[w_idx];
    for (size_t j = 0; j < t; ++j) {
        uint32_t a = gen_add32(p, data[j], data[t + j]);
        uint32_t b = gen_mul32(p, gen_sub32(p, data[j], data[t + j]), w1);
        data[j] = a;
        data[t + j] = b;
    }
}

void orc_plan32_fwd(const orc_plan32 *plan, uint32_t *buf) { /* src/prime32.rs:709-755 */
    uint32_t p = plan->p;
    size_t n = plan->n;
    if (p < ((uint32_t)1 << 30))
        fwd_depth_first_scalar32(p, buf, n, plan->twid, plan->twid_shoup, 0, 0, fwd_bfly30, fwd_last_bfly30);
    else if (p < ((uint32_t)1 << 31))
        fwd_depth_first_scalar32(p, buf, n, plan->twid, plan->twid_shoup, 0, 0, fwd_bfly31, fwd_last_bfly31);
    else
        gen_fwd_depth_first32(buf, n, p, plan->twid, 0, 0);
}

void orc_plan32_inv(const orc_plan32 *plan, uint32_t *buf) { /* src/prime32.rs:762-808 */
    uint32_t p = plan->p;
    size_t n = plan->n;
    if (p < ((uint32_t)1 << 30))
        inv_depth_first_scalar32(p, buf, n, plan->inv_twid, plan->inv_twid_shoup, 0, 0, inv_bfly30,
                                 inv_last_bfly30);
    else if (p < ((uint32_t)1 << 31))
        inv_depth_first_scalar32(p, buf, n, plan->inv_twid, plan->inv_twid_shoup, 0, 0, inv_bfly31, inv_bfly31);
    else
        gen_inv_depth_first32(buf, n, p, plan->inv_twid, 0, 0);
}

void orc_plan32_mul_assign_normalize(const orc_plan32 *plan, uint32_t *lhs, const uint32_t *rhs, size_t len) {
    uint32_t p = plan->p;
    if (p < ((uint32_t)1 << 31)) { /* src/prime32.rs:383-408 */
        uint32_t big_q_m1 = plan->big_q - 1;
        for (size_t i = 0; i < len; ++i) {
            uint64_t d = (uint64_t)lhs[i] * rhs[i];
            uint32_t c1 = (uint32_t)(d >> big_q_m1);
            uint32_t c3 = (uint32_t)(((uint64_t)c1 * plan->p_barrett) >> 32);
            uint32_t prod = (uint32_t)d - p * c3;
            uint32_t shoup_q = (uint32_t)(((uint64_t)prod * plan->n_inv_mod_p_shoup) >> 32);
            uint32_t t = prod * plan->n_inv_mod_p - shoup_q * p;
            lhs[i] = min32(t, t - p);
        }
    } else { /* :852-863 */
        for (size_t i = 0; i < len; ++i) {
            uint32_t prod = gen_mul32(p, lhs[i], rhs[i]);
            lhs[i] = gen_mul32(p, prod, plan->n_inv_mod_p);
        }
    }
}

void orc_plan32_normalize(const orc_plan32 *plan, uint32_t *values, size_t len) {
    uint32_t p = plan->p;
    if (p < ((uint32_t)1 << 31)) { /* src/prime32.rs:477-488 */
        for (size_t i = 0; i < len; ++i) {
            uint32_t val = values[i];
            uint32_t shoup_q = (uint32_t)(((uint64_t)val * plan->n_inv_mod_p_shoup) >> 32);
            uint32_t t = val * plan->n_inv_mod_p - shoup_q * p;
            values[i] = min32(t, t - p);
        }
    } else { /* :890-898 */
        for (size_t i = 0; i < len; ++i) values[i] = gen_mul32(p, values[i], plan->n_inv_mod_p);
    }
}

void orc_plan32_mul_accumulate(const orc_plan32 *plan, uint32_t *acc, const uint32_t *lhs, const uint32_t *rhs,
                               size_t len) {
    uint32_t p = plan->p;
    if (p < ((uint32_t)1 << 31)) { /* src/prime32.rs:575-598 */
        uint32_t big_q_m1 = plan->big_q - 1;
        for (size_t i = 0; i < len; ++i) {
            uint64_t d = (uint64_t)lhs[i] * rhs[i];
            uint32_t c1 = (uint32_t)(d >> big_q_m1);
            uint32_t c3 = (uint32_t)(((uint64_t)c1 * plan->p_barrett) >> 32);
            uint32_t prod = (uint32_t)d - p * c3;
            prod = min32(prod, prod - p);
            uint32_t acc_ = prod + acc[i];
            acc[i] = min32(acc_, acc_ - p);
        }
    } else { /* :917-925 */
        for (size_t i = 0; i < len; ++i) {
            uint32_t prod = gen_mul32(p, lhs[i], rhs[i]);
            acc[i] = gen_add32(p, acc[i], prod);
        }
    }
}

/* ========================================================================= */
/* CRT constants: src/lib.rs:447-652 (const fn there; computed on first use)  */
/* ========================================================================= */
static const uint32_t P32[10] = {
    0x3F5A0001u, 0x3F5D0001u, 0x3F760001u, 0x3F820001u, 0x3FAC0001u, /* src/lib.rs:453-457 */
    0x3FAF0001u, 0x3FB10001u, 0x3FBB0001u, 0x3FDE0001u, 0x3FFC0001u, /* src/lib.rs:458-462 */
};
static const uint64_t P52[6] = {
    0x3FFFFFE770001ull, 0x3FFFFFEB90001ull, 0x3FFFFFEC80001ull, /* src/lib.rs:601-603 */
    0x3FFFFFF8B0001ull, 0x3FFFFFFB80001ull, 0x3FFFFFFC70001ull, /* src/lib.rs:604-606 */
};
uint32_t orc_primes32_p(int i) { return P32[i]; }
uint64_t orc_primes52_p(int i) { return P52[i]; }

static inline uint32_t inv_mod32(uint32_t m, uint32_t x) { return orc_exp_mod32(m, x, m - 2); } /* :491-493 */
static inline uint64_t shoup64(uint64_t m, uint64_t w) { return (uint64_t)((((u128)w) << 64) / m); } /* :508-510 */
static inline uint64_t inv_mod64p(uint64_t m, uint64_t x) { return orc_exp_mod64(m, x, m - 2); }     /* :624-626 */

/* src/native32.rs:21-25 (also used as native64::mul_mod32) */
static inline uint32_t n_mul_mod32(uint32_t p, uint32_t a, uint32_t b) { return (uint32_t)(((uint64_t)a * b) % p); }
/* src/native64.rs:36-41 */
static inline uint64_t n_mul_mod64(uint64_t p_neg, uint64_t a, uint64_t b, uint64_t b_shoup) {
    uint64_t q = (uint64_t)(((u128)a * b_shoup) >> 64);
    uint64_t r = a * b + p_neg * q;
    return min64(r, r + p_neg);
}

uint32_t orc_reconstruct_32bit_01(uint32_t mod_p0, uint32_t mod_p1) { /* src/native_binary32.rs:22-41 */
    const uint32_t P0 = P32[0], P1 = P32[1];
    uint32_t P0_INV_MOD_P1 = inv_mod32(P1, P0);
    uint32_t v0 = mod_p0;
    uint32_t v1 = n_mul_mod32(P1, P0_INV_MOD_P1, 2 * P1 + mod_p1 - v0);
    int sign = v1 > (P1 / 2);
    uint32_t _0 = P0, _01 = _0 * P1;
    uint32_t pos = v0 + v1 * _0;
    uint32_t neg = pos - _01;
    return sign ? neg : pos;
}

static void digits_012(uint32_t mod_p0, uint32_t mod_p1, uint32_t mod_p2, uint32_t *v0o, uint32_t *v1o,
                       uint32_t *v2o) { /* shared head of src/native32.rs:31-38 and src/native_binary64.rs:36-43 */
    const uint32_t P0 = P32[0], P1 = P32[1], P2 = P32[2];
    uint32_t P0_INV_MOD_P1 = inv_mod32(P1, P0);
    uint32_t P01_INV_MOD_P2 = inv_mod32(P2, n_mul_mod32(P2, P0, P1));
    uint32_t v0 = mod_p0;
    uint32_t v1 = n_mul_mod32(P1, P0_INV_MOD_P1, 2 * P1 + mod_p1 - v0);
    uint32_t v2 = n_mul_mod32(P2, P01_INV_MOD_P2, 2 * P2 + mod_p2 - (v0 + n_mul_mod32(P2, P0, v1)));
    *v0o = v0;
    *v1o = v1;
    *v2o = v2;
}

uint32_t orc_reconstruct_32bit_012_u32(uint32_t m0, uint32_t m1, uint32_t m2) { /* src/native32.rs:28-56 */
    uint32_t v0, v1, v2;
    digits_012(m0, m1, m2, &v0, &v1, &v2);
    int sign = v2 > (P32[2] / 2);
    uint32_t _0 = P32[0], _01 = _0 * P32[1], _012 = _01 * P32[2];
    uint32_t pos = v0 + v1 * _0 + v2 * _01;
    uint32_t neg = pos - _012;
    return sign ? neg : pos;
}

uint64_t orc_reconstruct_32bit_012_u64(uint32_t m0, uint32_t m1, uint32_t m2) { /* src/native_binary64.rs:33-61 */
    uint32_t v0, v1, v2;
    digits_012(m0, m1, m2, &v0, &v1, &v2);
    int sign = v2 > (P32[2] / 2);
    uint64_t _0 = P32[0], _01 = _0 * (uint64_t)P32[1], _012 = _01 * (uint64_t)P32[2];
    uint64_t pos = (uint64_t)v0 + (uint64_t)v1 * _0 + (uint64_t)v2 * _01;
    uint64_t neg = pos - _012;
    return sign ? neg : pos;
}

/* shared head of src/native64.rs:98-125 and src/native_binary128.rs:20-45 */
static void digits_0_12_34(const uint32_t m[5], uint64_t *v0o, uint64_t *v12o, uint64_t *v34o, uint64_t *p12o,
                           uint64_t *p34o) {
    const uint32_t P0 = P32[0], P1 = P32[1], P2 = P32[2], P3 = P32[3], P4 = P32[4];
    uint32_t P1_INV_MOD_P2 = inv_mod32(P2, P1);
    uint32_t P3_INV_MOD_P4 = inv_mod32(P4, P3);
    uint64_t P12 = (uint64_t)P1 * P2, P34 = (uint64_t)P3 * P4; /* src/lib.rs:539-540 */
    /* src/lib.rs:541-551: inverse through Euler's theorem, phi(P12) = (P1-1)(P2-1) */
    uint64_t P0_INV_MOD_P12 = orc_exp_mod64(P12, P0, ((uint64_t)P1 - 1) * ((uint64_t)P2 - 1) - 1);
    uint64_t P0_INV_MOD_P12_SHOUP = shoup64(P12, P0_INV_MOD_P12);
    uint64_t P0_MOD_P34_SHOUP = shoup64(P34, P0);
    uint64_t P012_INV_MOD_P34 =
        orc_exp_mod64(P34, orc_mul_mod64(P34, P0, P12), ((uint64_t)P3 - 1) * ((uint64_t)P4 - 1) - 1);
    uint64_t P012_INV_MOD_P34_SHOUP = shoup64(P34, P012_INV_MOD_P34);

    uint64_t mod_p12, mod_p34;
    {
        uint32_t v1 = m[1];
        uint32_t v2 = n_mul_mod32(P2, P1_INV_MOD_P2, 2 * P2 + m[2] - v1);
        mod_p12 = (uint64_t)v1 + ((uint64_t)v2 * P1);
    }
    {
        uint32_t v3 = m[3];
        uint32_t v4 = n_mul_mod32(P4, P3_INV_MOD_P4, 2 * P4 + m[4] - v3);
        mod_p34 = (uint64_t)v3 + ((uint64_t)v4 * P3);
    }
    uint64_t v0 = m[0];
    uint64_t v12 = n_mul_mod64((uint64_t)0 - P12, 2 * P12 + mod_p12 - v0, P0_INV_MOD_P12, P0_INV_MOD_P12_SHOUP);
    uint64_t v34 = n_mul_mod64((uint64_t)0 - P34,
                               2 * P34 + mod_p34 - (v0 + n_mul_mod64((uint64_t)0 - P34, v12, P0, P0_MOD_P34_SHOUP)),
                               P012_INV_MOD_P34, P012_INV_MOD_P34_SHOUP);
    *v0o = v0;
    *v12o = v12;
    *v34o = v34;
    *p12o = P12;
    *p34o = P34;
}

uint64_t orc_reconstruct_32bit_01234_v2_u64(const uint32_t m[5]) { /* src/native64.rs:91-141 */
    uint64_t v0, v12, v34, P12, P34;
    digits_0_12_34(m, &v0, &v12, &v34, &P12, &P34);
    int sign = v34 > (P34 / 2);
    uint64_t _0 = P32[0], _012 = _0 * P12, _01234 = _012 * P34;
    uint64_t pos = v0 + v12 * _0 + v34 * _012;
    uint64_t neg = pos - _01234;
    return sign ? neg : pos;
}

orc_u128 orc_reconstruct_32bit_01234_v2_u128(const uint32_t m[5]) { /* src/native_binary128.rs:13-63 */
    uint64_t v0, v12, v34, P12, P34;
    digits_0_12_34(m, &v0, &v12, &v34, &P12, &P34);
    int sign = v34 > (P34 / 2);
    u128 _0 = P32[0], _012 = _0 * (u128)P12, _01234 = _012 * (u128)P34;
    u128 pos = (u128)v0 + (u128)v12 * _0 + (u128)v34 * _012;
    u128 neg = pos - _01234;
    return sign ? neg : pos;
}

orc_u128 orc_reconstruct_32bit_0123456789_v2(const uint32_t m[10]) { /* src/native128.rs:20-118 */
    const uint32_t *P = P32;
    uint64_t modp[5]; /* mod_p01, mod_p23, mod_p45, mod_p67, mod_p89 : :34-58 */
    for (int k = 0; k < 5; ++k) {
        uint32_t pa = P[2 * k], pb = P[2 * k + 1];
        uint32_t inv = inv_mod32(pb, pa); /* P0_INV_MOD_P1, P2_INV_MOD_P3, ... src/lib.rs:554-561 */
        uint32_t va = m[2 * k];
        uint32_t vb = n_mul_mod32(pb, inv, 2 * pb + m[2 * k + 1] - va);
        modp[k] = (uint64_t)va + ((uint64_t)vb * pa);
    }
    uint64_t P01 = (uint64_t)P[0] * P[1], P23 = (uint64_t)P[2] * P[3], P45 = (uint64_t)P[4] * P[5],
             P67 = (uint64_t)P[6] * P[7], P89 = (uint64_t)P[8] * P[9]; /* src/lib.rs:563-567 */
    /* src/lib.rs:569-595 */
    uint64_t P01_MOD_P45_SHOUP = shoup64(P45, P01), P01_MOD_P67_SHOUP = shoup64(P67, P01),
             P01_MOD_P89_SHOUP = shoup64(P89, P01);
    uint64_t P23_MOD_P67_SHOUP = shoup64(P67, P23), P23_MOD_P89_SHOUP = shoup64(P89, P23);
    uint64_t P45_MOD_P89_SHOUP = shoup64(P89, P45);
    uint64_t P01_INV_MOD_P23 = orc_exp_mod64(P23, P01, ((uint64_t)P[2] - 1) * ((uint64_t)P[3] - 1) - 1);
    uint64_t P0123_INV_MOD_P45 =
        orc_exp_mod64(P45, orc_mul_mod64(P45, P01, P23), ((uint64_t)P[4] - 1) * ((uint64_t)P[5] - 1) - 1);
    uint64_t P012345_INV_MOD_P67 =
        orc_exp_mod64(P67, orc_mul_mod64(P67, orc_mul_mod64(P67, P01, P23), P45),
                      ((uint64_t)P[6] - 1) * ((uint64_t)P[7] - 1) - 1);
    uint64_t P01234567_INV_MOD_P89 =
        orc_exp_mod64(P89, orc_mul_mod64(P89, orc_mul_mod64(P89, orc_mul_mod64(P89, P01, P23), P45), P67),
                      ((uint64_t)P[8] - 1) * ((uint64_t)P[9] - 1) - 1);
    uint64_t nP23 = (uint64_t)0 - P23, nP45 = (uint64_t)0 - P45, nP67 = (uint64_t)0 - P67, nP89 = (uint64_t)0 - P89;

    uint64_t v01 = modp[0];
    uint64_t v23 = n_mul_mod64(nP23, 2 * P23 + modp[1] - v01, P01_INV_MOD_P23, shoup64(P23, P01_INV_MOD_P23));
    uint64_t v45 = n_mul_mod64(nP45, 2 * P45 + modp[2] - (v01 + n_mul_mod64(nP45, v23, P01, P01_MOD_P45_SHOUP)),
                               P0123_INV_MOD_P45, shoup64(P45, P0123_INV_MOD_P45));
    uint64_t v67 = n_mul_mod64(
        nP67,
        2 * P67 + modp[3] -
            (v01 + n_mul_mod64(nP67, v23 + n_mul_mod64(nP67, v45, P23, P23_MOD_P67_SHOUP), P01, P01_MOD_P67_SHOUP)),
        P012345_INV_MOD_P67, shoup64(P67, P012345_INV_MOD_P67));
    uint64_t v89 = n_mul_mod64(
        nP89,
        2 * P89 + modp[4] -
            (v01 + n_mul_mod64(nP89,
                               v23 + n_mul_mod64(nP89, v45 + n_mul_mod64(nP89, v67, P45, P45_MOD_P89_SHOUP), P23,
                                                 P23_MOD_P89_SHOUP),
                               P01, P01_MOD_P89_SHOUP)),
        P01234567_INV_MOD_P89, shoup64(P89, P01234567_INV_MOD_P89));

    int sign = v89 > (P89 / 2);
    u128 P0123 = (u128)P01 * (u128)P23;   /* src/lib.rs:592 */
    u128 P012345 = P0123 * (u128)P45;     /* :593 */
    u128 P01234567 = P012345 * (u128)P67; /* :594 */
    u128 P0123456789 = P01234567 * (u128)P89; /* :595 (wrapping) */
    u128 pos = (u128)v01 + (u128)v23 * (u128)P01 + (u128)v45 * P0123 + (u128)v67 * P012345 +
               (u128)v89 * P01234567;
    u128 neg = pos - P0123456789;
    return sign ? neg : pos;
}

/* ---- 52-bit plans.  The reference implements these only with AVX-512 IFMA
 * (mul_mod52_avx512, src/native32.rs:96-107: Shoup product on 52-bit lanes followed by one
 * conditional subtraction => canonical a*b mod p).  Restated as exact arithmetic. ---- */
static inline uint64_t mul_mod52(uint64_t p, uint64_t a, uint64_t b) { return (uint64_t)(((u128)a * b) % p); }

uint32_t orc_reconstruct_52bit_0(uint64_t mod_p0) { /* src/native_binary32.rs:111-125 */
    uint64_t P0 = P52[0];
    int sign = mod_p0 > (P0 / 2);
    uint64_t pos = mod_p0, neg = pos - P0;
    return (uint32_t)(sign ? neg : pos);
}
static uint64_t rec52_01(uint64_t mod_p0, uint64_t mod_p1) { /* src/native32.rs:223-253, native_binary64.rs:230-260 */
    uint64_t P0 = P52[0], P1 = P52[1];
    uint64_t P0_INV_MOD_P1 = inv_mod64p(P1, P0);
    uint64_t v0 = mod_p0;
    uint64_t v1 = mul_mod52(P1, 2 * P1 + mod_p1 - v0, P0_INV_MOD_P1);
    int sign = v1 > (P1 / 2);
    uint64_t pos = v0 + v1 * P0;
    uint64_t neg = pos - P0 * P1;
    return sign ? neg : pos;
}
uint32_t orc_reconstruct_52bit_01_u32(uint64_t m0, uint64_t m1) { return (uint32_t)rec52_01(m0, m1); }
uint64_t orc_reconstruct_52bit_01_u64(uint64_t m0, uint64_t m1) { return rec52_01(m0, m1); }
uint64_t orc_reconstruct_52bit_012(uint64_t mod_p0, uint64_t mod_p1, uint64_t mod_p2) { /* src/native64.rs:770-829 */
    uint64_t P0 = P52[0], P1 = P52[1], P2 = P52[2];
    uint64_t P0_INV_MOD_P1 = inv_mod64p(P1, P0);
    uint64_t P01_INV_MOD_P2 = inv_mod64p(P2, mul_mod52(P2, P0, P1));
    uint64_t v0 = mod_p0;
    uint64_t v1 = mul_mod52(P1, 2 * P1 + mod_p1 - v0, P0_INV_MOD_P1);
    uint64_t v2 = mul_mod52(P2, 2 * P2 + mod_p2 - (v0 + mul_mod52(P2, v1, P0)), P01_INV_MOD_P2);
    int sign = v2 > (P2 / 2);
    uint64_t pos = v0 + v1 * P0 + v2 * (P0 * P1);
    uint64_t neg = pos - P0 * P1 * P2;
    return sign ? neg : pos;
}

/* ========================================================================= */
/* native plans                                                              */
/* ========================================================================= */
orc_native *orc_native_try_new(orc_native_kind kind, size_t n) {
    static const int NPR[10] = {3, 5, 10, 2, 3, 5, 2, 3, 1, 2};
    static const int WORD[10] = {4, 8, 16, 4, 8, 16, 4, 8, 4, 8};
    orc_native *pl = (orc_native *)calloc(1, sizeof(*pl));
    pl->kind = kind;
    pl->n = n;
    pl->nprimes = NPR[kind];
    pl->word = WORD[kind];
    pl->is52 = kind >= ORC_NATIVE32_PLAN52;
    pl->binary = (kind >= ORC_NATIVE_BINARY32_PLAN32 && kind <= ORC_NATIVE_BINARY128_PLAN32) ||
                 kind == ORC_NATIVE_BINARY32_PLAN52 || kind == ORC_NATIVE_BINARY64_PLAN52;
    for (int i = 0; i < pl->nprimes; ++i) {
        if (pl->is52) {
            pl->p64[i] = orc_plan64_try_new(n, P52[i], NULL);
            if (!pl->p64[i]) {
                orc_native_free(pl);
                return NULL;
            }
        } else {
            pl->p32[i] = orc_plan32_try_new(n, P32[i], NULL); /* `?` propagation: src/native64.rs:933-942 */
            if (!pl->p32[i]) {
                orc_native_free(pl);
                return NULL;
            }
        }
    }
    return pl;
}

void orc_native_free(orc_native *pl) {
    if (!pl) return;
    for (int i = 0; i < 10; ++i) orc_plan32_free(pl->p32[i]);
    for (int i = 0; i < 3; ++i) orc_plan64_free(pl->p64[i]);
    free(pl);
}

static inline u128 load_word(const void *v, int word, size_t i) {
    if (word == 4) return ((const uint32_t *)v)[i];
    if (word == 8) return ((const uint64_t *)v)[i];
    u128 x;
    memcpy(&x, (const char *)v + 16 * i, 16); /* u128 as Rust lays it out on x86-64: 16-byte little-endian */
    return x;
}
static inline void store_word(void *v, int word, size_t i, u128 x) {
    if (word == 4)
        ((uint32_t *)v)[i] = (uint32_t)x;
    else if (word == 8)
        ((uint64_t *)v)[i] = (uint64_t)x;
    else
        memcpy((char *)v + 16 * i, &x, 16);
}

static void native_split(const orc_native *pl, const void *value, void *const *res, int binary) {
    size_t n = pl->n;
    for (size_t i = 0; i < n; ++i) {
        u128 v = load_word(value, pl->word, i);
        for (int k = 0; k < pl->nprimes; ++k) {
            if (pl->is52) {
                /* src/native64.rs:1113-1119 (value % P_i); src/native32.rs:447-452 and
                 * src/native_binary32.rs:272-279 copy the u32 (always < P_i);
                 * fwd_binary: src/native_binary64.rs:481-487 plain copy */
                uint64_t r = (binary || pl->word == 4) ? (uint64_t)v : (uint64_t)(v % P52[k]);
                ((uint64_t *)res[k])[i] = r;
            } else {
                /* src/native64.rs:980-993 (value % P_i);
                 * fwd_binary: src/native_binary64.rs:379-385 `*value as u32` */
                uint32_t r = binary ? (uint32_t)v : (uint32_t)(v % P32[k]);
                ((uint32_t *)res[k])[i] = r;
            }
        }
    }
    for (int k = 0; k < pl->nprimes; ++k) {
        if (pl->is52)
            orc_plan64_fwd(pl->p64[k], (uint64_t *)res[k]);
        else
            orc_plan32_fwd(pl->p32[k], (uint32_t *)res[k]);
    }
}

void orc_native_fwd(const orc_native *pl, const void *value, void *const *res) { native_split(pl, value, res, 0); }
void orc_native_fwd_binary(const orc_native *pl, const void *value, void *const *res) {
    native_split(pl, value, res, 1);
}

void orc_native_inv(const orc_native *pl, void *value, void *const *res) {
    size_t n = pl->n;
    for (int k = 0; k < pl->nprimes; ++k) {
        if (pl->is52)
            orc_plan64_inv(pl->p64[k], (uint64_t *)res[k]);
        else
            orc_plan32_inv(pl->p32[k], (uint32_t *)res[k]);
    }
    for (size_t i = 0; i < n; ++i) {
        uint32_t m[10];
        uint64_t m64[3];
        for (int k = 0; k < pl->nprimes; ++k) {
            if (pl->is52)
                m64[k] = ((uint64_t *)res[k])[i];
            else
                m[k] = ((uint32_t *)res[k])[i];
        }
        u128 out = 0;
        switch (pl->kind) {
        case ORC_NATIVE32_PLAN32: out = orc_reconstruct_32bit_012_u32(m[0], m[1], m[2]); break;
        case ORC_NATIVE64_PLAN32: out = orc_reconstruct_32bit_01234_v2_u64(m); break;
        case ORC_NATIVE128_PLAN32: out = orc_reconstruct_32bit_0123456789_v2(m); break;
        case ORC_NATIVE_BINARY32_PLAN32: out = orc_reconstruct_32bit_01(m[0], m[1]); break;
        case ORC_NATIVE_BINARY64_PLAN32: out = orc_reconstruct_32bit_012_u64(m[0], m[1], m[2]); break;
        case ORC_NATIVE_BINARY128_PLAN32: out = orc_reconstruct_32bit_01234_v2_u128(m); break;
        case ORC_NATIVE32_PLAN52: out = orc_reconstruct_52bit_01_u32(m64[0], m64[1]); break;
        case ORC_NATIVE64_PLAN52: out = orc_reconstruct_52bit_012(m64[0], m64[1], m64[2]); break;
        case ORC_NATIVE_BINARY32_PLAN52: out = orc_reconstruct_52bit_0(m64[0]); break;
        case ORC_NATIVE_BINARY64_PLAN52: out = orc_reconstruct_52bit_01_u64(m64[0], m64[1]); break;
        }
        store_word(value, pl->word, i, out);
    }
}

void orc_native_negacyclic_polymul(const orc_native *pl, void *prod, const void *lhs, const void *rhs) {
    /* src/native64.rs:1042-1069: 2k temporaries, fwd lhs, fwd (or fwd_binary) rhs, k pointwise, inv */
    size_t n = pl->n;
    size_t rb = pl->is52 ? 8 : 4;
    void *l[10], *r[10];
    for (int k = 0; k < pl->nprimes; ++k) {
        l[k] = xaligned(n * rb);
        r[k] = xaligned(n * rb);
    }
    orc_native_fwd(pl, lhs, l);
    if (pl->binary)
        orc_native_fwd_binary(pl, rhs, r);
    else
        orc_native_fwd(pl, rhs, r);
    for (int k = 0; k < pl->nprimes; ++k) {
        if (pl->is52)
            orc_plan64_mul_assign_normalize(pl->p64[k], (uint64_t *)l[k], (const uint64_t *)r[k], n);
        else
            orc_plan32_mul_assign_normalize(pl->p32[k], (uint32_t *)l[k], (const uint32_t *)r[k], n);
    }
    orc_native_inv(pl, prod, l);
    for (int k = 0; k < pl->nprimes; ++k) {
        free(l[k]);
        free(r[k]);
    }
}

/* ========================================================================= */
/* product::Plan  (src/product.rs)                                           */
/* ========================================================================= */
static uint32_t p_sub_mod_u32(uint32_t m, uint32_t a, uint32_t b) { return a >= b ? a - b : a - b + m; } /* :76-82 */
static uint64_t p_sub_mod_u64(uint64_t m, uint64_t a, uint64_t b) { return a >= b ? a - b : a - b + m; } /* :67-73 */
static uint64_t p_add_mod_u64(uint64_t m, uint64_t a, uint64_t b) { /* :85-92 */
    uint64_t sum = a + b;
    int overflow = sum < a;
    return (sum >= m || overflow) ? sum - m : sum;
}
static uint32_t p_add_mod_u32(uint32_t m, uint32_t a, uint32_t b) { /* :107-114 */
    uint32_t sum = a + b;
    int overflow = sum < a;
    return (sum >= m || overflow) ? sum - m : sum;
}
static uint32_t modular_inv_u32(uint32_t modulus, uint32_t n) { /* :22-42 extended Euclid */
    uint32_t old_r = n % modulus, r = modulus, old_s = 1, s = 0;
    while (r != 0) {
        uint32_t q = old_r / r;
        uint32_t nr = old_r - q * r;
        old_r = r;
        r = nr;
        uint32_t ns = p_sub_mod_u32(modulus, old_s, mul_mod32(modulus, q, s));
        old_s = s;
        s = ns;
    }
    return old_s;
}
static uint64_t modular_inv_u64(uint64_t modulus, uint64_t n) { /* :44-64 */
    uint64_t old_r = n % modulus, r = modulus, old_s = 1, s = 0;
    while (r != 0) {
        uint64_t q = old_r / r;
        uint64_t nr = old_r - q * r;
        old_r = r;
        r = nr;
        uint64_t ns = p_sub_mod_u64(modulus, old_s, orc_mul_mod64(modulus, q, s));
        old_s = s;
        s = ns;
    }
    return old_s;
}
static int cmp_u64(const void *a, const void *b) {
    uint64_t x = *(const uint64_t *)a, y = *(const uint64_t *)b;
    return x < y ? -1 : (x > y ? 1 : 0);
}

orc_product *orc_product_try_new(size_t n, uint64_t modulus, const uint64_t *factors, size_t nfactors) { /* :153-247 */
    if (n % 2 != 0 || nfactors > 16) return NULL;
    uint64_t primes[16];
    memcpy(primes, factors, nfactors * sizeof(uint64_t));
    qsort(primes, nfactors, sizeof(uint64_t), cmp_u64);
    uint64_t prev = 0;
    for (size_t i = 0; i < nfactors; ++i) { /* zeros / duplicates */
        if (primes[i] == prev) return NULL;
        prev = primes[i];
    }
    size_t start = 0;
    while (start < nfactors && primes[start] == 1) ++start;
    uint64_t prod = 1;
    for (size_t i = start; i < nfactors; ++i) { /* checked_mul */
        u128 w = (u128)prod * primes[i];
        if (w >> 64) return NULL;
        prod = (uint64_t)w;
    }
    if (prod != modulus) return NULL;
    orc_product *pl = (orc_product *)calloc(1, sizeof(*pl));
    pl->n = n;
    pl->modulus = modulus;
    size_t len = nfactors - start;
    for (size_t i = 0; i < len; ++i) pl->primes[i] = primes[start + i];
    for (size_t i = 0; i < len; ++i) {
        if (pl->primes[i] < ((uint64_t)1 << 32)) {
            orc_plan32 *sub = orc_plan32_try_new(n, (uint32_t)pl->primes[i], NULL);
            if (!sub) {
                orc_product_free(pl);
                return NULL;
            }
            pl->p32[pl->n32++] = sub;
        } else {
            orc_plan64 *sub = orc_plan64_try_new(n, pl->primes[i], NULL);
            if (!sub) {
                orc_product_free(pl);
                return NULL;
            }
            pl->p64[pl->n64++] = sub;
        }
    }
    size_t off = 0;
    for (size_t j = 0; j < len; ++j)
        for (size_t i = 0; i < j; ++i)
            pl->modular_inverses[off++] = (j < (size_t)pl->n32)
                                              ? (uint64_t)modular_inv_u32((uint32_t)pl->primes[j], (uint32_t)pl->primes[i])
                                              : modular_inv_u64(pl->primes[j], pl->primes[i]);
    return pl;
}
void orc_product_free(orc_product *pl) {
    if (!pl) return;
    for (int i = 0; i < 16; ++i) {
        orc_plan32_free(pl->p32[i]);
        orc_plan64_free(pl->p64[i]);
    }
    free(pl);
}
size_t orc_product_modular_inverses(const orc_product *pl, uint64_t *out) {
    size_t len = (size_t)(pl->n32 + pl->n64), cnt = len ? len * (len - 1) / 2 : 0;
    memcpy(out, pl->modular_inverses, cnt * sizeof(uint64_t));
    return cnt;
}
size_t orc_product_ntt_domain_len(const orc_product *pl) { return (pl->n / 2) * (size_t)pl->n32 + pl->n * (size_t)pl->n64; }

void orc_product_fwd(const orc_product *pl, uint64_t *ntt, const uint64_t *standard, int bounded, uint64_t bound) {
    size_t n = pl->n;
    uint32_t *ntt32 = (uint32_t *)ntt;
    uint64_t *ntt64 = ntt + (n / 2) * (size_t)pl->n32;
    if (pl->n32 == 0 && pl->n64 == 1) { /* :282-286 */
        memcpy(ntt64, standard, n * 8);
        orc_plan64_fwd(pl->p64[0], ntt64);
        return;
    }
    if (pl->n32 == 1 && pl->n64 == 0) { /* :287-293: truncation, no % */
        for (size_t i = 0; i < n; ++i) ntt32[i] = (uint32_t)standard[i];
        orc_plan32_fwd(pl->p32[0], ntt32);
        return;
    }
    if (pl->n32 == 2 && pl->n64 == 0) { /* :295-335 */
        uint32_t *ntt0 = ntt32, *ntt1 = ntt32 + n;
        uint32_t p0 = pl->p32[0]->p, p1 = pl->p32[1]->p;
        uint64_t p = pl->modulus;
        uint32_t p_u32 = (uint32_t)p;
        if (bounded && bound < (uint64_t)p0 && bound < (uint64_t)p1) {
            for (size_t i = 0; i < n; ++i) {
                int positive = standard[i] < p / 2;
                uint32_t s = (uint32_t)standard[i];
                uint32_t complement = p_u32 - s;
                ntt0[i] = positive ? s : p0 - complement;
                ntt1[i] = positive ? s : p1 - complement;
            }
        } else {
            for (size_t i = 0; i < n; ++i) {
                ntt0[i] = (uint32_t)(standard[i] % p0);
                ntt1[i] = (uint32_t)(standard[i] % p1);
            }
        }
        orc_plan32_fwd(pl->p32[0], ntt0);
        orc_plan32_fwd(pl->p32[1], ntt1);
        return;
    }
    for (int k = 0; k < pl->n32; ++k) { /* :337-345 */
        uint32_t *dst = ntt32 + (size_t)k * n;
        for (size_t i = 0; i < n; ++i) dst[i] = (uint32_t)(standard[i] % pl->p32[k]->p);
        orc_plan32_fwd(pl->p32[k], dst);
    }
    for (int k = 0; k < pl->n64; ++k) { /* :347-355 */
        uint64_t *dst = ntt64 + (size_t)k * n;
        for (size_t i = 0; i < n; ++i) dst[i] = standard[i] % pl->p64[k]->p;
        orc_plan64_fwd(pl->p64[k], dst);
    }
}

void orc_product_inv(const orc_product *pl, uint64_t *standard, uint64_t *ntt, int accumulate) {
    size_t n = pl->n;
    uint32_t *ntt32 = (uint32_t *)ntt;
    uint64_t *ntt64 = ntt + (n / 2) * (size_t)pl->n32;
    for (int k = 0; k < pl->n32; ++k) orc_plan32_inv(pl->p32[k], ntt32 + (size_t)k * n);
    for (int k = 0; k < pl->n64; ++k) orc_plan64_inv(pl->p64[k], ntt64 + (size_t)k * n);
    uint64_t p = pl->modulus;
    if (pl->n32 == 0 && pl->n64 == 0) { /* :378-384 */
        if (!accumulate) memset(standard, 0, n * 8);
        return;
    }
    if (pl->n32 == 0 && pl->n64 == 1) { /* :386-398 */
        for (size_t i = 0; i < n; ++i) standard[i] = accumulate ? p_add_mod_u64(pl->p64[0]->p, standard[i], ntt64[i]) : ntt64[i];
        return;
    }
    if (pl->n32 == 1 && pl->n64 == 0) { /* :399-415 */
        for (size_t i = 0; i < n; ++i)
            standard[i] = accumulate ? (uint64_t)p_add_mod_u32(pl->p32[0]->p, (uint32_t)standard[i], ntt32[i]) : (uint64_t)ntt32[i];
        return;
    }
    /* general Garner (Knuth 4.3.2), src/product.rs:791-879; the u32x2 special case :419-789 computes the same
     * digits v0 = u0, v1 = (u1 - v0) * p0^-1 mod p1 (Shoup or exact product: equal values) */
    size_t len = (size_t)(pl->n32 + pl->n64);
    for (size_t idx = 0; idx < n; ++idx) {
        uint64_t v[16];
        size_t off = 0;
        for (size_t j = 0; j < len; ++j) {
            uint64_t pj = pl->primes[j];
            uint64_t x = (j < (size_t)pl->n32) ? (uint64_t)ntt32[j * n + idx] : ntt64[(j - (size_t)pl->n32) * n + idx];
            for (size_t i = 0; i < j; ++i) {
                uint64_t inv = pl->modular_inverses[off + i];
                if (j < (size_t)pl->n32) {
                    uint32_t diff = p_sub_mod_u32((uint32_t)pj, (uint32_t)x, (uint32_t)v[i]);
                    x = mul_mod32((uint32_t)pj, diff, (uint32_t)inv);
                } else {
                    uint64_t diff = p_sub_mod_u64(pj, x, v[i]);
                    x = orc_mul_mod64(pj, diff, inv);
                }
            }
            off += j;
            v[j] = x;
        }
        uint64_t acc = 0;
        for (size_t j = len; j-- > 0;) { /* :861-872 */
            acc *= pl->primes[j];
            acc += v[j];
        }
        standard[idx] = accumulate ? p_add_mod_u64(p, standard[idx], acc) : acc;
    }
}

void orc_product_mul_assign_normalize(const orc_product *pl, uint64_t *lhs, const uint64_t *rhs) {
    size_t n = pl->n;
    for (int k = 0; k < pl->n32; ++k)
        orc_plan32_mul_assign_normalize(pl->p32[k], (uint32_t *)lhs + (size_t)k * n, (const uint32_t *)rhs + (size_t)k * n, n);
    size_t o = (n / 2) * (size_t)pl->n32;
    for (int k = 0; k < pl->n64; ++k)
        orc_plan64_mul_assign_normalize(pl->p64[k], lhs + o + (size_t)k * n, rhs + o + (size_t)k * n, n);
}
void orc_product_normalize(const orc_product *pl, uint64_t *values) {
    size_t n = pl->n;
    for (int k = 0; k < pl->n32; ++k) orc_plan32_normalize(pl->p32[k], (uint32_t *)values + (size_t)k * n, n);
    size_t o = (n / 2) * (size_t)pl->n32;
    for (int k = 0; k < pl->n64; ++k) orc_plan64_normalize(pl->p64[k], values + o + (size_t)k * n, n);
}
void orc_product_mul_accumulate(const orc_product *pl, uint64_t *acc, const uint64_t *lhs, const uint64_t *rhs) {
    size_t n = pl->n;
    for (int k = 0; k < pl->n32; ++k)
        orc_plan32_mul_accumulate(pl->p32[k], (uint32_t *)acc + (size_t)k * n, (const uint32_t *)lhs + (size_t)k * n,
                                  (const uint32_t *)rhs + (size_t)k * n, n);
    size_t o = (n / 2) * (size_t)pl->n32;
    for (int k = 0; k < pl->n64; ++k)
        orc_plan64_mul_accumulate(pl->p64[k], acc + o + (size_t)k * n, lhs + o + (size_t)k * n, rhs + o + (size_t)k * n, n);
}

/* ========================================================================= */
/* AVX-512 restatement of the 62-bit-class engine (CPU baseline only)        */
/*   src/prime64/shoup.rs:10-156 (fwd_breadth_first_avx512), :712-870 (inv),  */
/*   butterflies src/prime64/less_than_62bit.rs:7-57 (fwd) and the inverse    */
/*   pair; 64x64->128 emulation as src/lib.rs:171-199 (four vpmuludq).        */
/* Built only where the compiler targets AVX-512F+DQ (the `native` build on  */
/* the machine that times the baseline); same values as the scalar engine --  */
/* tests/test_oracle_golden.py compares them when the build has it.          */
/* ========================================================================= */
#if defined(__AVX512F__) && defined(__AVX512DQ__)
#include <immintrin.h>
int orc_avx512_available(void) { return __builtin_cpu_supports("avx512f") && __builtin_cpu_supports("avx512dq"); }

typedef __m512i v8;
static inline v8 v_small_mod(v8 m, v8 x) { return _mm512_min_epu64(x, _mm512_sub_epi64(x, m)); }
/* high 64 bits of the lane-wise 64x64 product: schoolbook on 32-bit halves */
static inline v8 v_mulhi(v8 a, v8 b) {
    const v8 mask = _mm512_set1_epi64(0xFFFFFFFFll);
    v8 ah = _mm512_srli_epi64(a, 32), bh = _mm512_srli_epi64(b, 32);
    v8 ll = _mm512_mul_epu32(a, b), lh = _mm512_mul_epu32(a, bh), hl = _mm512_mul_epu32(ah, b), hh = _mm512_mul_epu32(ah, bh);
    v8 mid = _mm512_add_epi64(lh, _mm512_srli_epi64(ll, 32));              /* < 2^64 */
    v8 mid2 = _mm512_add_epi64(hl, _mm512_and_si512(mid, mask));           /* < 2^64 */
    return _mm512_add_epi64(_mm512_add_epi64(hh, _mm512_srli_epi64(mid, 32)), _mm512_srli_epi64(mid2, 32));
}
static inline v8 v_shoup(v8 y, v8 w, v8 ws, v8 neg_p) { /* y*w - hi(y*ws)*p  (wrapping) */
    return _mm512_add_epi64(_mm512_mullo_epi64(y, w), _mm512_mullo_epi64(v_mulhi(y, ws), neg_p));
}
typedef struct { v8 p, neg_p, two_p; } vmod;
static inline void v_fwd_bfly(v8 *z0, v8 *z1, v8 w, v8 ws, const vmod *M, int last) {
    v8 x = v_small_mod(M->two_p, *z0);
    if (last) x = v_small_mod(M->p, x);
    v8 t = v_shoup(*z1, w, ws, M->neg_p);
    if (last) {
        t = v_small_mod(M->p, t);
        *z0 = v_small_mod(M->p, _mm512_add_epi64(x, t));
        *z1 = v_small_mod(M->p, _mm512_add_epi64(_mm512_sub_epi64(x, t), M->p));
    } else {
        *z0 = _mm512_add_epi64(x, t);
        *z1 = _mm512_add_epi64(_mm512_sub_epi64(x, t), M->two_p);
    }
}
static inline void v_inv_bfly(v8 *z0, v8 *z1, v8 w, v8 ws, const vmod *M, int last) {
    v8 y0 = v_small_mod(M->two_p, _mm512_add_epi64(*z0, *z1));
    if (last) y0 = v_small_mod(M->p, y0);
    v8 t = _mm512_add_epi64(_mm512_sub_epi64(*z0, *z1), M->two_p);
    v8 y1 = v_shoup(t, w, ws, M->neg_p);
    if (last) y1 = v_small_mod(M->p, y1);
    *z0 = y0;
    *z1 = y1;
}
/* in-vector stages: 16 consecutive coefficients (vectors A, B) regrouped so that lane i of Z0 and Z1 form a
 * butterfly of half-distance T in {4, 2, 1}; W gathers the matching twiddles.  Index tables for vpermt2q. */
static const long long SPLIT0[3][8] = {{0, 1, 2, 3, 8, 9, 10, 11}, {0, 1, 4, 5, 8, 9, 12, 13}, {0, 2, 4, 6, 8, 10, 12, 14}};
static const long long SPLIT1[3][8] = {{4, 5, 6, 7, 12, 13, 14, 15}, {2, 3, 6, 7, 10, 11, 14, 15}, {1, 3, 5, 7, 9, 11, 13, 15}};
static const long long JOIN0[3][8] = {{0, 1, 2, 3, 8, 9, 10, 11}, {0, 1, 8, 9, 2, 3, 10, 11}, {0, 8, 1, 9, 2, 10, 3, 11}};
static const long long JOIN1[3][8] = {{4, 5, 6, 7, 12, 13, 14, 15}, {4, 5, 12, 13, 6, 7, 14, 15}, {4, 12, 5, 13, 6, 14, 7, 15}};
static const long long WSEL[3][8] = {{0, 0, 0, 0, 1, 1, 1, 1}, {0, 0, 1, 1, 2, 2, 3, 3}, {0, 1, 2, 3, 4, 5, 6, 7}};
static inline void v_small_stage(uint64_t *data, size_t n, const uint64_t *w, const uint64_t *ws, int k /*0:T=4,1:T=2,2:T=1*/,
                                 const vmod *M, int inv, int last) {
    const v8 s0 = _mm512_loadu_si512(SPLIT0[k]), s1 = _mm512_loadu_si512(SPLIT1[k]);
    const v8 j0 = _mm512_loadu_si512(JOIN0[k]), j1 = _mm512_loadu_si512(JOIN1[k]);
    const v8 wsel = _mm512_loadu_si512(WSEL[k]);
    const size_t wstep = (size_t)2 << k; /* twiddles consumed per 16 coefficients */
    for (size_t i = 0, wi = 0; i < n; i += 16, wi += wstep) {
        v8 a = _mm512_loadu_si512(data + i), b = _mm512_loadu_si512(data + i + 8);
        v8 z0 = _mm512_permutex2var_epi64(a, s0, b), z1 = _mm512_permutex2var_epi64(a, s1, b);
        /* 8 twiddles are loaded, the first wstep are used (the tables are n long and wi + 8 <= n holds) */
        v8 wv = _mm512_permutexvar_epi64(wsel, _mm512_loadu_si512(w + wi));
        v8 wsv = _mm512_permutexvar_epi64(wsel, _mm512_loadu_si512(ws + wi));
        if (inv)
            v_inv_bfly(&z0, &z1, wv, wsv, M, last);
        else
            v_fwd_bfly(&z0, &z1, wv, wsv, M, last);
        _mm512_storeu_si512(data + i, _mm512_permutex2var_epi64(z0, j0, z1));
        _mm512_storeu_si512(data + i + 8, _mm512_permutex2var_epi64(z0, j1, z1));
    }
}
static int avx512_eligible(const orc_plan64 *pl) { return pl->twid_shoup && pl->p < ((uint64_t)1 << 62) && pl->n >= 16; }

void orc_plan64_fwd_avx512(const orc_plan64 *pl, uint64_t *data) {
    if (!avx512_eligible(pl)) {
        orc_plan64_fwd(pl, data);
        return;
    }
    const size_t n = pl->n;
    const vmod M = {_mm512_set1_epi64((long long)pl->p), _mm512_set1_epi64((long long)(0 - pl->p)),
                    _mm512_set1_epi64((long long)(2 * pl->p))};
    size_t t = n, m = 1;
    while (m < n / 8) { /* vector stages: t >= 8 */
        t /= 2;
        for (size_t i = 0; i < m; ++i) {
            const v8 w = _mm512_set1_epi64((long long)pl->twid[m + i]), ws = _mm512_set1_epi64((long long)pl->twid_shoup[m + i]);
            uint64_t *z0 = data + 2 * i * t, *z1 = z0 + t;
            for (size_t j = 0; j < t; j += 8) {
                v8 a = _mm512_loadu_si512(z0 + j), b = _mm512_loadu_si512(z1 + j);
                v_fwd_bfly(&a, &b, w, ws, &M, 0);
                _mm512_storeu_si512(z0 + j, a);
                _mm512_storeu_si512(z1 + j, b);
            }
        }
        m *= 2;
    }
    v_small_stage(data, n, pl->twid + n / 8, pl->twid_shoup + n / 8, 0, &M, 0, 0);
    v_small_stage(data, n, pl->twid + n / 4, pl->twid_shoup + n / 4, 1, &M, 0, 0);
    v_small_stage(data, n, pl->twid + n / 2, pl->twid_shoup + n / 2, 2, &M, 0, 1);
}
void orc_plan64_inv_avx512(const orc_plan64 *pl, uint64_t *data) {
    if (!avx512_eligible(pl)) {
        orc_plan64_inv(pl, data);
        return;
    }
    const size_t n = pl->n;
    const vmod M = {_mm512_set1_epi64((long long)pl->p), _mm512_set1_epi64((long long)(0 - pl->p)),
                    _mm512_set1_epi64((long long)(2 * pl->p))};
    v_small_stage(data, n, pl->inv_twid + n / 2, pl->inv_twid_shoup + n / 2, 2, &M, 1, 0);
    v_small_stage(data, n, pl->inv_twid + n / 4, pl->inv_twid_shoup + n / 4, 1, &M, 1, 0);
    v_small_stage(data, n, pl->inv_twid + n / 8, pl->inv_twid_shoup + n / 8, 0, &M, 1, 0);
    size_t t = 8, m = n / 16;
    while (m >= 1) { /* vector stages: t >= 8 */
        const int last = m == 1;
        for (size_t i = 0; i < m; ++i) {
            const v8 w = _mm512_set1_epi64((long long)pl->inv_twid[m + i]), ws = _mm512_set1_epi64((long long)pl->inv_twid_shoup[m + i]);
            uint64_t *z0 = data + 2 * i * t, *z1 = z0 + t;
            for (size_t j = 0; j < t; j += 8) {
                v8 a = _mm512_loadu_si512(z0 + j), b = _mm512_loadu_si512(z1 + j);
                v_inv_bfly(&a, &b, w, ws, &M, last);
                _mm512_storeu_si512(z0 + j, a);
                _mm512_storeu_si512(z1 + j, b);
            }
        }
        t *= 2;
        m /= 2;
    }
}
#else
int orc_avx512_available(void) { return 0; }
void orc_plan64_fwd_avx512(const orc_plan64 *pl, uint64_t *data) { orc_plan64_fwd(pl, data); }
void orc_plan64_inv_avx512(const orc_plan64 *pl, uint64_t *data) { orc_plan64_inv(pl, data); }
#endif

/* ========================================================================= */
/* schoolbook negacyclic convolution: src/prime64.rs:1143-1182               */
/* ========================================================================= */
static inline uint64_t t_add64(uint64_t p, uint64_t a, uint64_t b) {
    uint64_t neg_b = p - b; /* wrapping_sub; p == 0 -> wrapping arithmetic */
    return (a >= neg_b) ? a - neg_b : a + b;
}
static inline uint64_t t_sub64(uint64_t p, uint64_t a, uint64_t b) {
    uint64_t neg_b = p - b;
    return (a >= b) ? a - b : a + neg_b;
}
static inline uint64_t t_mul64(uint64_t p, uint64_t a, uint64_t b) {
    u128 wide = (u128)a * b;
    return p == 0 ? (uint64_t)wide : (uint64_t)(wide % p);
}
void orc_negacyclic_convolution64(size_t n, uint64_t p, const uint64_t *lhs, const uint64_t *rhs, uint64_t *out) {
    uint64_t *full = (uint64_t *)calloc(2 * n, 8);
    for (size_t i = 0; i < n; ++i)
        for (size_t j = 0; j < n; ++j) full[i + j] = t_add64(p, full[i + j], t_mul64(p, lhs[i], rhs[j]));
    for (size_t i = 0; i < n; ++i) out[i] = t_sub64(p, full[i], full[i + n]);
    free(full);
}
void orc_negacyclic_convolution32(size_t n, uint32_t p, const uint32_t *lhs, const uint32_t *rhs, uint32_t *out) {
    /* src/prime32.rs tests module: same shape on u32 (p == 0 -> wrapping mod 2^32) */
    uint32_t *full = (uint32_t *)calloc(2 * n, 4);
    for (size_t i = 0; i < n; ++i)
        for (size_t j = 0; j < n; ++j) {
            uint64_t wide = (uint64_t)lhs[i] * rhs[j];
            uint32_t m = p == 0 ? (uint32_t)wide : (uint32_t)(wide % p);
            uint32_t a = full[i + j], neg_b = p - m;
            full[i + j] = (a >= neg_b) ? a - neg_b : a + m;
        }
    for (size_t i = 0; i < n; ++i) {
        uint32_t a = full[i], b = full[i + n], neg_b = p - b;
        out[i] = (a >= b) ? a - b : a + neg_b;
    }
    free(full);
}
void orc_negacyclic_convolution128(size_t n, const orc_u128 *lhs, const orc_u128 *rhs, orc_u128 *out) {
    /* src/native128.rs:359-372: wrapping u128 */
    u128 *full = (u128 *)calloc(2 * n, 16);
    for (size_t i = 0; i < n; ++i)
        for (size_t j = 0; j < n; ++j) full[i + j] += lhs[i] * rhs[j];
    for (size_t i = 0; i < n; ++i) out[i] = full[i] - full[i + n];
    free(full);
}

/* ========================================================================= */
/* synthetic inputs (shared definition with the GPU fill kernel)             */
/* ========================================================================= */
uint64_t orc_splitmix64(uint64_t x) {
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}
void orc_fill_uniform_u64(uint64_t *dst, size_t count, uint64_t bound, uint64_t seed) {
    for (size_t i = 0; i < count; ++i) {
        uint64_t r = orc_splitmix64(seed + i);
        dst[i] = bound ? (uint64_t)(((u128)r * bound) >> 64) : r;
    }
}
void orc_fill_uniform_u32(uint32_t *dst, size_t count, uint32_t bound, uint64_t seed) {
    for (size_t i = 0; i < count; ++i) {
        uint64_t r = orc_splitmix64(seed + i);
        dst[i] = bound ? (uint32_t)(((r >> 32) * (uint64_t)bound) >> 32) : (uint32_t)(r >> 32);
    }
}

/* ========================================================================= */
/* batched drivers                                                           */
/* ========================================================================= */
typedef struct {
    int op; /* 0 fwd64 1 inv64 2 mul64 3 fwd32 4 inv32 5 native polymul */
    const void *plan;
    void *a;
    const void *b;
    const void *c;
    size_t begin, end;
} job_t;

static void *job_main(void *arg) {
    job_t *j = (job_t *)arg;
    for (size_t i = j->begin; i < j->end; ++i) {
        switch (j->op) {
        case 0: {
            const orc_plan64 *pl = (const orc_plan64 *)j->plan;
            orc_plan64_fwd(pl, (uint64_t *)j->a + i * pl->n);
        } break;
        case 1: {
            const orc_plan64 *pl = (const orc_plan64 *)j->plan;
            orc_plan64_inv(pl, (uint64_t *)j->a + i * pl->n);
        } break;
        case 2: {
            const orc_plan64 *pl = (const orc_plan64 *)j->plan;
            orc_plan64_mul_assign_normalize(pl, (uint64_t *)j->a + i * pl->n, (const uint64_t *)j->b + i * pl->n,
                                            pl->n);
        } break;
        case 3: {
            const orc_plan32 *pl = (const orc_plan32 *)j->plan;
            orc_plan32_fwd(pl, (uint32_t *)j->a + i * pl->n);
        } break;
        case 4: {
            const orc_plan32 *pl = (const orc_plan32 *)j->plan;
            orc_plan32_inv(pl, (uint32_t *)j->a + i * pl->n);
        } break;
        case 6: {
            const orc_plan64 *pl = (const orc_plan64 *)j->plan;
            orc_plan64_fwd_avx512(pl, (uint64_t *)j->a + i * pl->n);
        } break;
        case 7: {
            const orc_plan64 *pl = (const orc_plan64 *)j->plan;
            orc_plan64_inv_avx512(pl, (uint64_t *)j->a + i * pl->n);
        } break;
        case 5: {
            const orc_native *pl = (const orc_native *)j->plan;
            size_t stride = pl->n * (size_t)pl->word;
            orc_native_negacyclic_polymul(pl, (char *)j->a + i * stride, (const char *)j->b + i * stride,
                                          (const char *)j->c + i * stride);
        } break;
        }
    }
    return NULL;
}

static double now_s(void) {
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

static double run_batch(int op, const void *plan, void *a, const void *b, const void *c, size_t batch, int nthreads) {
    if (nthreads < 1) nthreads = 1;
    if ((size_t)nthreads > batch) nthreads = batch ? (int)batch : 1;
    job_t *jobs = (job_t *)calloc((size_t)nthreads, sizeof(job_t));
    pthread_t *th = (pthread_t *)calloc((size_t)nthreads, sizeof(pthread_t));
    double t0 = now_s();
    for (int t = 0; t < nthreads; ++t) {
        jobs[t] = (job_t){op, plan, a, b, c, batch * (size_t)t / (size_t)nthreads,
                          batch * (size_t)(t + 1) / (size_t)nthreads};
        if (nthreads == 1)
            job_main(&jobs[t]);
        else
            pthread_create(&th[t], NULL, job_main, &jobs[t]);
    }
    if (nthreads > 1)
        for (int t = 0; t < nthreads; ++t) pthread_join(th[t], NULL);
    double t1 = now_s();
    free(jobs);
    free(th);
    return t1 - t0;
}

double orc_plan64_fwd_batch(const orc_plan64 *plan, uint64_t *bufs, size_t batch, int nthreads) {
    return run_batch(0, plan, bufs, NULL, NULL, batch, nthreads);
}
double orc_plan64_inv_batch(const orc_plan64 *plan, uint64_t *bufs, size_t batch, int nthreads) {
    return run_batch(1, plan, bufs, NULL, NULL, batch, nthreads);
}
double orc_plan64_mul_assign_normalize_batch(const orc_plan64 *plan, uint64_t *lhs, const uint64_t *rhs,
                                             size_t batch, int nthreads) {
    return run_batch(2, plan, lhs, rhs, NULL, batch, nthreads);
}
double orc_plan64_fwd_avx512_batch(const orc_plan64 *plan, uint64_t *bufs, size_t batch, int nthreads) {
    return run_batch(6, plan, bufs, NULL, NULL, batch, nthreads);
}
double orc_plan64_inv_avx512_batch(const orc_plan64 *plan, uint64_t *bufs, size_t batch, int nthreads) {
    return run_batch(7, plan, bufs, NULL, NULL, batch, nthreads);
}
/* CPU-baseline driver: every thread runs `reps` x (fwd, inv) over its contiguous share of the batch, so that the
 * timed region is transform work and not thread start-up (a 1024-point transform takes about a microsecond).
 * Returns the elapsed seconds; the number of transforms done is 2 * reps * batch. */
typedef struct {
    const orc_plan64 *plan;
    uint64_t *bufs;
    size_t lo, hi;
    int reps, avx512;
} loop_job_t;
static void *loop_job_main(void *arg) {
    const loop_job_t *j = (const loop_job_t *)arg;
    const size_t n = j->plan->n;
    for (int r = 0; r < j->reps; ++r)
        for (size_t i = j->lo; i < j->hi; ++i) {
            uint64_t *b = j->bufs + i * n;
            if (j->avx512) {
                orc_plan64_fwd_avx512(j->plan, b);
                orc_plan64_inv_avx512(j->plan, b);
            } else {
                orc_plan64_fwd(j->plan, b);
                orc_plan64_inv(j->plan, b);
            }
        }
    return NULL;
}
double orc_plan64_fwd_inv_loop(const orc_plan64 *plan, uint64_t *bufs, size_t batch, int nthreads, int reps, int avx512) {
    if (nthreads < 1) nthreads = 1;
    if ((size_t)nthreads > batch) nthreads = batch ? (int)batch : 1;
    loop_job_t *jobs = (loop_job_t *)calloc((size_t)nthreads, sizeof(loop_job_t));
    pthread_t *th = (pthread_t *)calloc((size_t)nthreads, sizeof(pthread_t));
    double t0 = now_s();
    for (int t = 0; t < nthreads; ++t) {
        jobs[t] = (loop_job_t){plan, bufs, batch * (size_t)t / (size_t)nthreads, batch * (size_t)(t + 1) / (size_t)nthreads,
                               reps, avx512};
        if (nthreads == 1)
            loop_job_main(&jobs[t]);
        else
            pthread_create(&th[t], NULL, loop_job_main, &jobs[t]);
    }
    if (nthreads > 1)
        for (int t = 0; t < nthreads; ++t) pthread_join(th[t], NULL);
    double t1 = now_s();
    free(jobs);
    free(th);
    return t1 - t0;
}
double orc_plan32_fwd_batch(const orc_plan32 *plan, uint32_t *bufs, size_t batch, int nthreads) {
    return run_batch(3, plan, bufs, NULL, NULL, batch, nthreads);
}
double orc_plan32_inv_batch(const orc_plan32 *plan, uint32_t *bufs, size_t batch, int nthreads) {
    return run_batch(4, plan, bufs, NULL, NULL, batch, nthreads);
}
double orc_native_negacyclic_polymul_batch(const orc_native *plan, void *prod, const void *lhs, const void *rhs,
                                           size_t batch, int nthreads) {
    return run_batch(5, plan, prod, lhs, rhs, batch, nthreads);
}
