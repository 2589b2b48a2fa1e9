/*
 * cntt_oracle.h -- CPU ORACLE (TEST INFRASTRUCTURE ONLY, NOT THE PRODUCT).
 *
 * Plain-C restatement of the scalar CPU path of zama-ai/concrete-ntt v0.2.0
 * (reference @ 2024-11-15).  Only tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg may load this library; the shipped library
 * (concrete-ntt_amd/csrc) never links or calls it.
 *
 * Parity pin: the reference is Rust and no Rust toolchain exists in this
 * image, so oracle/_ref cannot be built.  The restatement is pinned by
 *   (1) the known answers of SURVEY.md 8(c) (independent big-int derivation),
 *   (2) tests/golden/ fixtures (JSON) produced by tests/golden/gen_golden.py, a second,
 *       independent pure-Python big-int restatement, and
 *   (3) the reference's own property suite (README.md:41-57,
 *       src/prime64.rs:1211-1267, src/native64.rs:1176-1243, ...).
 * What that pins and what it does not: everything the reference's tests state as literals or
 * properties (prime search answers, inv(fwd(x)) = n x, products equal to the schoolbook wrapping
 * convolution, CRT round trips) is pinned.  The raw NTT-DOMAIN words (the bytes fwd() leaves) are
 * PARITY UNPINNED against the reference itself: it holds no such vectors and cannot be run here;
 * they are fixed only by the agreement of the independent restatements (1) and (2).
 *
 * Every function cites the reference file:line it follows (paths relative to
 * /root/reference/).
 */
#ifndef CNTT_ORACLE_H
#define CNTT_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef unsigned __int128 orc_u128;

/* ---- number theory (src/prime.rs, src/roots.rs, src/lib.rs) ---- */
size_t orc_bit_rev(uint32_t nbits, size_t i);                       /* src/lib.rs:118-121 */
uint64_t orc_mul_mod64(uint64_t p, uint64_t x, uint64_t y);         /* src/prime.rs:8-10 */
uint64_t orc_exp_mod64(uint64_t p, uint64_t base, uint64_t pow);    /* src/prime.rs:31-48 */
uint32_t orc_exp_mod32(uint32_t p, uint32_t base, uint32_t pow);    /* src/prime.rs:12-29 */
int orc_is_prime64(uint64_t n);                                     /* src/prime.rs:76-126 */
/* returns 1 and writes *out on Some, 0 on None.  src/prime.rs:130-180 */
int orc_largest_prime_in_arithmetic_progression64(uint64_t factor, uint64_t offset, uint64_t lo,
                                                  uint64_t hi, uint64_t *out);
int orc_get_z64(uint64_t p, uint64_t *z);                           /* src/roots.rs:17-28 */
/* src/roots.rs:31-66 ; returns 1 on Some */
int orc_sqrt_mod_ex64(uint64_t p, uint64_t q, uint64_t s, uint64_t z, uint64_t n, uint64_t *out);
/* src/roots.rs:68-91 ; returns 1 on Some */
int orc_find_primitive_root64(uint64_t p, uint64_t degree, uint64_t *root);

/* ---- prime64::Plan (src/prime64.rs:221-236) ---- */
typedef struct orc_plan64 {
    size_t n;
    uint64_t p;
    uint64_t *twid, *twid_shoup, *inv_twid, *inv_twid_shoup; /* shoup tables NULL when p >= 2^63 */
    uint64_t p_barrett, big_q, n_inv_mod_p, n_inv_mod_p_shoup;
} orc_plan64;

/* src/prime64.rs:704-771.  NULL == None.  p <= 1 aborts in the reference (Div64::new panics,
 * src/fastdiv.rs:99); here it returns NULL and sets *panicked=1 when panicked != NULL. */
orc_plan64 *orc_plan64_try_new(size_t n, uint64_t p, int *panicked);
void orc_plan64_free(orc_plan64 *plan);
void orc_plan64_fwd(const orc_plan64 *plan, uint64_t *buf);                      /* src/prime64.rs:794-865 */
void orc_plan64_inv(const orc_plan64 *plan, uint64_t *buf);                      /* src/prime64.rs:872-943 */
void orc_plan64_mul_assign_normalize(const orc_plan64 *plan, uint64_t *lhs, const uint64_t *rhs,
                                     size_t len);                                /* :947-1033 */
void orc_plan64_normalize(const orc_plan64 *plan, uint64_t *values, size_t len); /* :1037-1082 */
void orc_plan64_mul_accumulate(const orc_plan64 *plan, uint64_t *acc, const uint64_t *lhs,
                               const uint64_t *rhs, size_t len);                 /* :1085-1128 */

/* AVX-512 restatement of the 62-bit-class transforms (src/prime64/shoup.rs:10-156, :712-870): present in builds that
 * target AVX-512F+DQ (make native on such a host), scalar otherwise; CPU-baseline use only. */
int orc_avx512_available(void);
void orc_plan64_fwd_avx512(const orc_plan64 *plan, uint64_t *buf);
void orc_plan64_inv_avx512(const orc_plan64 *plan, uint64_t *buf);

/* ---- prime32::Plan (src/prime32.rs:601-616) ---- */
typedef struct orc_plan32 {
    size_t n;
    uint32_t p;
    uint32_t *twid, *twid_shoup, *inv_twid, *inv_twid_shoup; /* shoup tables NULL when p >= 2^31 */
    uint32_t p_barrett, big_q, n_inv_mod_p, n_inv_mod_p_shoup;
} orc_plan32;

orc_plan32 *orc_plan32_try_new(size_t n, uint32_t p, int *panicked);             /* src/prime32.rs:630-686 */
void orc_plan32_free(orc_plan32 *plan);
void orc_plan32_fwd(const orc_plan32 *plan, uint32_t *buf);                      /* :709-755 */
void orc_plan32_inv(const orc_plan32 *plan, uint32_t *buf);                      /* :762-808 */
void orc_plan32_mul_assign_normalize(const orc_plan32 *plan, uint32_t *lhs, const uint32_t *rhs,
                                     size_t len);                                /* :812-864 */
void orc_plan32_normalize(const orc_plan32 *plan, uint32_t *values, size_t len); /* :868-899 */
void orc_plan32_mul_accumulate(const orc_plan32 *plan, uint32_t *acc, const uint32_t *lhs,
                               const uint32_t *rhs, size_t len);                 /* :902-927 */

/* ---- CRT constants (src/lib.rs:447-652) ---- */
uint32_t orc_primes32_p(int i); /* P0..P9 */
uint64_t orc_primes52_p(int i); /* P0..P5 */

/* scalar CRT reconstructions */
uint32_t orc_reconstruct_32bit_01(uint32_t m0, uint32_t m1);                      /* src/native_binary32.rs:22-41 */
uint32_t orc_reconstruct_32bit_012_u32(uint32_t m0, uint32_t m1, uint32_t m2);    /* src/native32.rs:28-56 */
uint64_t orc_reconstruct_32bit_012_u64(uint32_t m0, uint32_t m1, uint32_t m2);    /* src/native_binary64.rs:33-61 */
uint64_t orc_reconstruct_32bit_01234_v2_u64(const uint32_t m[5]);                 /* src/native64.rs:91-141 */
orc_u128 orc_reconstruct_32bit_01234_v2_u128(const uint32_t m[5]);                /* src/native_binary128.rs:13-63 */
orc_u128 orc_reconstruct_32bit_0123456789_v2(const uint32_t m[10]);               /* src/native128.rs:20-118 */
uint32_t orc_reconstruct_52bit_0(uint64_t m0);                                    /* src/native_binary32.rs:111-125 */
uint32_t orc_reconstruct_52bit_01_u32(uint64_t m0, uint64_t m1);                  /* src/native32.rs:223-253 */
uint64_t orc_reconstruct_52bit_01_u64(uint64_t m0, uint64_t m1);                  /* src/native_binary64.rs:230-260 */
uint64_t orc_reconstruct_52bit_012(uint64_t m0, uint64_t m1, uint64_t m2);        /* src/native64.rs:770-829 */

/* ---- native plans ----
 * kind selects the reference type; word = bytes per coefficient (4, 8, 16);
 * nprimes = number of prime32 (Plan32) or prime64 (Plan52) sub-plans. */
typedef enum {
    ORC_NATIVE32_PLAN32 = 0,         /* src/native32.rs:8-12       3 x u32 primes, u32 words   */
    ORC_NATIVE64_PLAN32 = 1,         /* src/native64.rs:16-22      5 x u32 primes, u64 words   */
    ORC_NATIVE128_PLAN32 = 2,        /* src/native128.rs:6-17     10 x u32 primes, u128 words  */
    ORC_NATIVE_BINARY32_PLAN32 = 3,  /* src/native_binary32.rs:11  2 x u32 primes, u32 words   */
    ORC_NATIVE_BINARY64_PLAN32 = 4,  /* src/native_binary64.rs:17  3 x u32 primes, u64 words   */
    ORC_NATIVE_BINARY128_PLAN32 = 5, /* src/native_binary128.rs:4  5 x u32 primes, u128 words  */
    ORC_NATIVE32_PLAN52 = 6,         /* src/native32.rs:19         2 x 50-bit primes           */
    ORC_NATIVE64_PLAN52 = 7,         /* src/native64.rs:29-34      3 x 50-bit primes           */
    ORC_NATIVE_BINARY32_PLAN52 = 8,  /* src/native_binary32.rs:19  1 x 50-bit prime            */
    ORC_NATIVE_BINARY64_PLAN52 = 9,  /* src/native_binary64.rs:29  2 x 50-bit primes           */
} orc_native_kind;

typedef struct orc_native {
    orc_native_kind kind;
    size_t n;
    int nprimes;
    int word;    /* bytes per coefficient: 4, 8 or 16 */
    int is52;    /* residues are u64 (prime64 plans) instead of u32 */
    int binary;  /* has fwd_binary / rhs is binary in negacyclic_polymul */
    orc_plan32 *p32[10];
    orc_plan64 *p64[3];
} orc_native;

orc_native *orc_native_try_new(orc_native_kind kind, size_t n);
void orc_native_free(orc_native *plan);
/* value: n words; residues: nprimes arrays of n u32 (or u64 when is52), passed as array of pointers.
 * src/native64.rs:971-999 and siblings. */
void orc_native_fwd(const orc_native *plan, const void *value, void *const *residues);
/* binary plans only; src/native_binary64.rs:372-389 and siblings */
void orc_native_fwd_binary(const orc_native *plan, const void *value, void *const *residues);
/* src/native64.rs:1001-1038: inverse NTT each residue buffer in place, then CRT into value */
void orc_native_inv(const orc_native *plan, void *value, void *const *residues);
/* src/native64.rs:1042-1069 */
void orc_native_negacyclic_polymul(const orc_native *plan, void *prod, const void *lhs, const void *rhs);

/* ---- product::Plan (src/product.rs:139-967): NTT modulo a product of distinct primes ----
 * NTT-domain layout of ONE polynomial (src/product.rs:261-270): n32 * n u32 residues (packed two per
 * u64 word), then n64 * n u64 residues; ntt_domain_len = (n/2)*n32 + n*n64 words. */
typedef struct orc_product {
    size_t n;
    uint64_t modulus;
    int n32, n64;               /* primes below / not below 2^32, ascending (src/product.rs:183-184) */
    uint64_t primes[16];
    orc_plan32 *p32[16];
    orc_plan64 *p64[16];
    uint64_t modular_inverses[120]; /* src/product.rs:207-229: for j, for i < j: p_i^-1 mod p_j */
} orc_product;
/* FwdMode: bounded < 0 -> Generic, else Bounded(bound) */
orc_product *orc_product_try_new(size_t n, uint64_t modulus, const uint64_t *factors, size_t nfactors); /* :153-247 */
void orc_product_free(orc_product *plan);
size_t orc_product_modular_inverses(const orc_product *plan, uint64_t *out); /* copies the field, returns its length */
size_t orc_product_ntt_domain_len(const orc_product *plan);                                            /* :261-270 */
void orc_product_fwd(const orc_product *plan, uint64_t *ntt, const uint64_t *standard, int bounded,
                     uint64_t bound);                                                                  /* :273-357 */
/* accumulate == 0 -> InvMode::Replace, else InvMode::Accumulate */
void orc_product_inv(const orc_product *plan, uint64_t *standard, uint64_t *ntt, int accumulate);      /* :360-879 */
void orc_product_mul_assign_normalize(const orc_product *plan, uint64_t *lhs, const uint64_t *rhs);    /* :885-913 */
void orc_product_normalize(const orc_product *plan, uint64_t *values);                                 /* :917-931 */
void orc_product_mul_accumulate(const orc_product *plan, uint64_t *acc, const uint64_t *lhs,
                                const uint64_t *rhs);                                                  /* :935-966 */

/* ---- the reference tests' own oracle: schoolbook negacyclic convolution ----
 * src/prime64.rs:1170-1182 (p == 0 means wrapping arithmetic mod 2^64) */
void orc_negacyclic_convolution64(size_t n, uint64_t p, const uint64_t *lhs, const uint64_t *rhs,
                                  uint64_t *out);
void orc_negacyclic_convolution32(size_t n, uint32_t p, const uint32_t *lhs, const uint32_t *rhs,
                                  uint32_t *out);                       /* src/prime32.rs tests */
void orc_negacyclic_convolution128(size_t n, const orc_u128 *lhs, const orc_u128 *rhs,
                                   orc_u128 *out);                      /* src/native128.rs:359-372 */

/* ---- synthetic inputs shared with the GPU generator: splitmix64 stream,
 * element i of a buffer = mulhi64(splitmix64(seed + i), bound) (bound == 0 -> full 64-bit) ---- */
uint64_t orc_splitmix64(uint64_t x);
void orc_fill_uniform_u64(uint64_t *dst, size_t count, uint64_t bound, uint64_t seed);
void orc_fill_uniform_u32(uint32_t *dst, size_t count, uint32_t bound, uint64_t seed);

/* ---- batched helpers (for the checker and for bench.py's cpu_baseline leg) ----
 * nthreads <= 1 -> single thread (the reference's own criterion harness is single-threaded,
 * benches/ntt.rs:94-105). Returns elapsed seconds of the transform loop only. */
double orc_plan64_fwd_batch(const orc_plan64 *plan, uint64_t *bufs, size_t batch, int nthreads);
double orc_plan64_inv_batch(const orc_plan64 *plan, uint64_t *bufs, size_t batch, int nthreads);
double orc_plan64_mul_assign_normalize_batch(const orc_plan64 *plan, uint64_t *lhs,
                                             const uint64_t *rhs, size_t batch, int nthreads);
double orc_plan64_fwd_avx512_batch(const orc_plan64 *plan, uint64_t *bufs, size_t batch, int nthreads);
double orc_plan64_inv_avx512_batch(const orc_plan64 *plan, uint64_t *bufs, size_t batch, int nthreads);
double orc_plan64_fwd_inv_loop(const orc_plan64 *plan, uint64_t *bufs, size_t batch, int nthreads, int reps, int avx512);
double orc_plan32_fwd_batch(const orc_plan32 *plan, uint32_t *bufs, size_t batch, int nthreads);
double orc_plan32_inv_batch(const orc_plan32 *plan, uint32_t *bufs, size_t batch, int nthreads);
double orc_native_negacyclic_polymul_batch(const orc_native *plan, void *prod, const void *lhs,
                                           const void *rhs, size_t batch, int nthreads);

#ifdef __cplusplus
}
#endif
#endif
