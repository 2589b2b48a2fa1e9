/*
 * cntt.h -- C ABI of libcntt_hip.so, the MI355X (gfx950) drop-in for the hot path of
 * zama-ai/concrete-ntt v0.2.0: Plan::try_new / fwd / inv / mul_assign_normalize / normalize /
 * mul_accumulate for prime32 / prime64, the native* / native_binary* polynomial products, and
 * product::Plan (the CRT-of-primes plan built on the prime plans).
 *
 * The reference has no FFI layer: its boundary is its public Rust API.  Each entry
 * point below replaces the Rust item cited next to it (paths relative to the reference tree); a Rust
 * shim re-creates the types on top (rust/ in this repo, and INTEGRATION.md).
 *
 * Conventions
 *   - plain pointers and sizes only; handles are opaque and immutable after creation
 *     (reference plans are Send + Sync: any thread may use one plan on distinct buffers).
 *   - status codes instead of Option / panic:
 *       CNTT_OK       success
 *       CNTT_NONE     try_new returned None (n too small, not a power of two, modulus not prime,
 *                     or no primitive 2n-th root: src/prime64.rs:709-713, src/prime32.rs:635-640)
 *       CNTT_EINVAL   the reference panics (modulus <= 1: src/fastdiv.rs:48,99) or a NULL argument
 *       CNTT_ELEN     the reference's assert_eq!(buf.len(), ntt_size) fails (src/prime64.rs:795,873)
 *       CNTT_EDEVICE  HIP error (no GPU, launch or copy failure) -- there is NO CPU fallback
 *       CNTT_ENOMEM   allocation failure
 *   - input contract of the transforms: coefficients are canonical, 0 <= x < modulus -- what the reference's tests feed
 *     (src/prime64.rs:1234-1252) and the only range on which its back ends agree with each other.  The
 *     kernels use it (the lazy classes skip the first stage's conditional subtraction); larger words give unspecified
 *     residues, as they do in the reference.  Every output is canonical.
 *   - host-slice calls (no _batch suffix) mirror the Rust methods one to one: data is copied to the
 *     current HIP device, transformed by the HIP kernels, and copied back, synchronously.
 *   - _batch calls are the measured path: `batch` polynomials stored back to back (polynomial b at
 *     base + b * n), in place, resident in device memory (CNTT_MEM_DEVICE) or in host memory
 *     (CNTT_MEM_HOST: staged through the device), enqueued on `stream` (a hipStream_t, NULL = default
 *     stream).  Device-resident calls return after enqueueing; they never synchronise.
 */
#ifndef CNTT_H
#define CNTT_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum cntt_status {
    CNTT_OK = 0,
    CNTT_NONE = 1,
    CNTT_EINVAL = 2,
    CNTT_ELEN = 3,
    CNTT_EDEVICE = 4,
    CNTT_ENOMEM = 5
} cntt_status_t;

typedef enum cntt_mem { CNTT_MEM_HOST = 0, CNTT_MEM_DEVICE = 1 } cntt_mem_t;

/* which table of a prime plan (src/prime64.rs:221-236, src/prime32.rs:601-616) */
typedef enum cntt_table {
    CNTT_TWID = 0,
    CNTT_TWID_SHOUP = 1,
    CNTT_INV_TWID = 2,
    CNTT_INV_TWID_SHOUP = 3
} cntt_table_t;

/* scalar fields of a prime plan (same struct for both widths; u32 values zero-extended) */
typedef struct cntt_plan_info {
    uint64_t ntt_size, modulus;
    uint64_t p_barrett, big_q, n_inv_mod_p, n_inv_mod_p_shoup; /* src/prime64.rs:752-756 */
    uint64_t root;      /* the primitive 2n-th root w chosen by src/roots.rs:68-91 */
    int32_t has_shoup;  /* 0 when modulus >= 2^(B-1): no Shoup tables (src/prime64.rs:729-736) */
    int32_t arith_class; /* device arithmetic class of the transforms: 0 lazy (p < 2^(B-2)), 1 strict, 2 generic,
                          3 / 4 double-precision FMA (64-bit words, p < 2^50 / p < 2^51), 5 pseudo-Mersenne 2^64 - c, c < 2^32,
                          6 double-precision FMA on 32-bit words (p >= 2^31, n <= 16384)  (3-6: sizes that live in LDS) */
} cntt_plan_info_t;

const char *cntt_last_error(void);      /* thread-local description of the last non-OK status */
int cntt_device_count(void);            /* number of visible HIP devices (0 if none / no driver) */
const char *cntt_version(void);

/* Contiguous partition of a batch of independent polynomials over `world` devices (the reference has no counterpart: it is
 * single-threaded, src/lib.rs): rank r owns [begin, end), remainders go to the low ranks.  Pure arithmetic, no device call. */
int cntt_shard_bounds(size_t batch, int world, int rank, size_t *begin, size_t *end);

/* ------------------------------------------------------------------------------------- */
/* TESTING ONLY -- kernel-selection overrides for A/B timing and device-vs-device parity   */
/* tests.  Results are bit-identical for every setting; only the kernel that runs differs. */
/* The library never reads the process environment.  No counterpart in the reference.      */
/*   key              default  meaning                                                      */
/*   "fp"                1     double-precision classes (u64 p < 2^51, u32 p >= 2^31); 0: integer butterflies.  Read at plan creation */
/*   "pm64"              1     fold-by-c class for p = 2^64 - c, c < 2^32; 0: Montgomery class.  Read at plan creation                 */
/*   "blk"               1     wave-block walk (u64 n = 4096..16384, u32 n = 8192..32768); 0: one polynomial per workgroup            */
/*   "mul32_blk"         1     fused product of 32-bit words, n = 16384 / 32768, on the walk                                          */
/*   "ext32_blk"         1     one-output mul_accumulate chain of 32-bit words on the walk                                            */
/*   "ext_one"           1     fused chain kernel for 32-bit words above n = 4096; 0: composed pipeline                               */
/*   "ext_split"        -1     split launches for 3 / 4 outputs where measured faster; 0 never; 1 always                              */
/*   "native_acc"        1     accumulating-CRT whole-product kernels; 0: the parked-tile kernels                                     */
/*   "product_fused"    -1     product::Plan: composed forward + fused inverse; 0 neither fused; 1 both                               */
/*   "plan52_via32"      1     negacyclic_polymul of the Plan52 native kinds through the Plan32 whole-product kernel; 0: 50-bit primes */
/* value -1 restores a switch's default; key "reset" restores all.  Process-wide, thread-safe (atomics); the two class switches are    */
/* recorded in the plan at creation (cntt_prime*_plan_info().arith_class reports the class in use), the others are read per call.      */
/* ------------------------------------------------------------------------------------- */
int cntt_debug_set(const char *key, int value);
int cntt_debug_get(const char *key, int *value);

/* ===================================================================================== */
/* prime64::Plan  (src/prime64.rs:221-236)                                                */
/* ===================================================================================== */
typedef struct cntt_plan64 cntt_plan64_t;

int cntt_prime64_plan_new(size_t polynomial_size, uint64_t modulus, cntt_plan64_t **out); /* Plan::try_new  src/prime64.rs:704-771 */
cntt_plan64_t *cntt_prime64_plan_clone(const cntt_plan64_t *plan);                        /* #[derive(Clone)] src/prime64.rs:221 */
void cntt_prime64_plan_free(cntt_plan64_t *plan);                                         /* Drop */
size_t cntt_prime64_ntt_size(const cntt_plan64_t *plan);                                  /* Plan::ntt_size src/prime64.rs:779-781 */
uint64_t cntt_prime64_modulus(const cntt_plan64_t *plan);                                 /* Plan::modulus  src/prime64.rs:785-787 */
int cntt_prime64_plan_info(const cntt_plan64_t *plan, cntt_plan_info_t *out);             /* private fields, for parity tests */
/* copies table `which` (ntt_size words) to out; CNTT_NONE if the plan has no such table */
int cntt_prime64_plan_table(const cntt_plan64_t *plan, cntt_table_t which, uint64_t *out, size_t len);

int cntt_prime64_fwd(const cntt_plan64_t *plan, uint64_t *buf, size_t len);               /* Plan::fwd src/prime64.rs:794-865 */
int cntt_prime64_inv(const cntt_plan64_t *plan, uint64_t *buf, size_t len);               /* Plan::inv src/prime64.rs:872-943 */
/* Plan::mul_assign_normalize src/prime64.rs:947-1033 -- like the scalar path, processes min(lhs_len, rhs_len) */
int cntt_prime64_mul_assign_normalize(const cntt_plan64_t *plan, uint64_t *lhs, size_t lhs_len, const uint64_t *rhs, size_t rhs_len);
int cntt_prime64_normalize(const cntt_plan64_t *plan, uint64_t *values, size_t len);      /* Plan::normalize src/prime64.rs:1037-1082 */
/* Plan::mul_accumulate src/prime64.rs:1085-1128 */
int cntt_prime64_mul_accumulate(const cntt_plan64_t *plan, uint64_t *acc, size_t acc_len, const uint64_t *lhs, size_t lhs_len, const uint64_t *rhs, size_t rhs_len);

int cntt_prime64_fwd_batch(const cntt_plan64_t *plan, uint64_t *bufs, size_t batch, cntt_mem_t where, void *stream);
int cntt_prime64_inv_batch(const cntt_plan64_t *plan, uint64_t *bufs, size_t batch, cntt_mem_t where, void *stream);
int cntt_prime64_mul_assign_normalize_batch(const cntt_plan64_t *plan, uint64_t *lhs, const uint64_t *rhs, size_t batch, cntt_mem_t where, void *stream);
int cntt_prime64_normalize_batch(const cntt_plan64_t *plan, uint64_t *values, size_t batch, cntt_mem_t where, void *stream);
int cntt_prime64_mul_accumulate_batch(const cntt_plan64_t *plan, uint64_t *acc, const uint64_t *lhs, const uint64_t *rhs, size_t batch, cntt_mem_t where, void *stream);
/* Fused lhs <- inv(mul_assign_normalize(fwd(lhs), rhs_ntt)): the composition a caller of the reference writes as
 * plan.fwd(a); plan.mul_assign_normalize(a, b_ntt); plan.inv(a)  (examples/mul_poly_prime.rs, src/prime64.rs:1254-1266),
 * i.e. the negacyclic product of lhs with the polynomial whose forward transform is rhs_ntt, in one pass over HBM
 * for n <= 32768 (u64: n >= 4096 except the Montgomery-class moduli), three launches otherwise.  Same
 * values as the three separate calls. */
int cntt_prime64_mul_ntt_batch(const cntt_plan64_t *plan, uint64_t *lhs, const uint64_t *rhs_ntt, size_t batch, cntt_mem_t where, void *stream);
/* Fused mul_accumulate chain, the composition a caller of the reference writes around the NTT as
 *     for j < nterms { plan.fwd(t_j); for o < nout { plan.mul_accumulate(acc_o, t_j, key[j][o]) } }  for o { plan.inv(acc_o) }
 * (Plan::fwd src/prime64.rs:794, Plan::mul_accumulate :1085-1128, Plan::inv :872) with acc_o starting at zero:
 *     out[b][o] = inv( sum_j fwd(terms[b][j]) (.) key_ntt[j][o] )            accumulate == 0
 *     out[b][o] = (out[b][o] + that) mod p                                   accumulate != 0
 * terms: batch x nterms polynomials (standard domain, element b's terms back to back); key_ntt: nterms x nout
 * polynomials in the NTT domain, shared by the whole batch; out: batch x nout polynomials.  Same values as the
 * separate calls (like them, the result carries the factor n of the unnormalised inverse).  One pass over HBM when
 * the twiddle images fit LDS (n <= 2048 for u64, 4096 for u32) and nout <= 4; composed launches otherwise. */
int cntt_prime64_external_product_batch(const cntt_plan64_t *plan, uint64_t *out, const uint64_t *terms, const uint64_t *key_ntt, size_t nterms, size_t nout, size_t batch, int accumulate, cntt_mem_t where, void *stream);

/* ===================================================================================== */
/* prime32::Plan  (src/prime32.rs:601-616)                                                */
/* ===================================================================================== */
typedef struct cntt_plan32 cntt_plan32_t;

int cntt_prime32_plan_new(size_t polynomial_size, uint32_t modulus, cntt_plan32_t **out); /* Plan::try_new  src/prime32.rs:630-686 */
cntt_plan32_t *cntt_prime32_plan_clone(const cntt_plan32_t *plan);
void cntt_prime32_plan_free(cntt_plan32_t *plan);
size_t cntt_prime32_ntt_size(const cntt_plan32_t *plan);                                  /* src/prime32.rs:694-696 */
uint32_t cntt_prime32_modulus(const cntt_plan32_t *plan);                                 /* src/prime32.rs:700-702 */
int cntt_prime32_plan_info(const cntt_plan32_t *plan, cntt_plan_info_t *out);
int cntt_prime32_plan_table(const cntt_plan32_t *plan, cntt_table_t which, uint32_t *out, size_t len);

int cntt_prime32_fwd(const cntt_plan32_t *plan, uint32_t *buf, size_t len);               /* Plan::fwd src/prime32.rs:709-755 */
int cntt_prime32_inv(const cntt_plan32_t *plan, uint32_t *buf, size_t len);               /* Plan::inv src/prime32.rs:762-808 */
int cntt_prime32_mul_assign_normalize(const cntt_plan32_t *plan, uint32_t *lhs, size_t lhs_len, const uint32_t *rhs, size_t rhs_len); /* src/prime32.rs:812-864 */
int cntt_prime32_normalize(const cntt_plan32_t *plan, uint32_t *values, size_t len);      /* src/prime32.rs:868-899 */
int cntt_prime32_mul_accumulate(const cntt_plan32_t *plan, uint32_t *acc, size_t acc_len, const uint32_t *lhs, size_t lhs_len, const uint32_t *rhs, size_t rhs_len); /* src/prime32.rs:902-927 */

int cntt_prime32_fwd_batch(const cntt_plan32_t *plan, uint32_t *bufs, size_t batch, cntt_mem_t where, void *stream);
int cntt_prime32_inv_batch(const cntt_plan32_t *plan, uint32_t *bufs, size_t batch, cntt_mem_t where, void *stream);
int cntt_prime32_mul_assign_normalize_batch(const cntt_plan32_t *plan, uint32_t *lhs, const uint32_t *rhs, size_t batch, cntt_mem_t where, void *stream);
int cntt_prime32_normalize_batch(const cntt_plan32_t *plan, uint32_t *values, size_t batch, cntt_mem_t where, void *stream);
int cntt_prime32_mul_accumulate_batch(const cntt_plan32_t *plan, uint32_t *acc, const uint32_t *lhs, const uint32_t *rhs, size_t batch, cntt_mem_t where, void *stream);
int cntt_prime32_mul_ntt_batch(const cntt_plan32_t *plan, uint32_t *lhs, const uint32_t *rhs_ntt, size_t batch, cntt_mem_t where, void *stream); /* see cntt_prime64_mul_ntt_batch */
int cntt_prime32_external_product_batch(const cntt_plan32_t *plan, uint32_t *out, const uint32_t *terms, const uint32_t *key_ntt, size_t nterms, size_t nout, size_t batch, int accumulate, cntt_mem_t where, void *stream); /* see cntt_prime64_external_product_batch */

/* ===================================================================================== */
/* native / native_binary plans                                                          */
/*                                                                                       */
/* One handle type; `kind` selects the reference type.  Coefficient words are u32 / u64 /  */
/* u128 (16-byte little-endian, as Rust lays u128 out on x86-64); residues are u32 for     */
/* Plan32 kinds and u64 for Plan52 kinds.  `residues` is an array of cntt_native_nprimes() */
/* pointers (the reference takes them as separate mod_p0.. arguments).                     */
/* ===================================================================================== */
typedef enum cntt_native_kind {
    CNTT_NATIVE32_PLAN32 = 0,         /* native32::Plan32          src/native32.rs:8-12,335-432    3 primes, u32  */
    CNTT_NATIVE64_PLAN32 = 1,         /* native64::Plan32          src/native64.rs:16-22,930-1070  5 primes, u64  */
    CNTT_NATIVE128_PLAN32 = 2,        /* native128::Plan32         src/native128.rs:6-17,120-349  10 primes, u128 */
    CNTT_NATIVE_BINARY32_PLAN32 = 3,  /* native_binary32::Plan32   src/native_binary32.rs:11,187-254 2 primes, u32 */
    CNTT_NATIVE_BINARY64_PLAN32 = 4,  /* native_binary64::Plan32   src/native_binary64.rs:17-21,342-445 3 primes, u64 */
    CNTT_NATIVE_BINARY128_PLAN32 = 5, /* native_binary128::Plan32  src/native_binary128.rs:4-10,65-197 5 primes, u128 */
    CNTT_NATIVE32_PLAN52 = 6,         /* native32::Plan52          src/native32.rs:19,434-496      2 x 50-bit     */
    CNTT_NATIVE64_PLAN52 = 7,         /* native64::Plan52          src/native64.rs:29-34,1074-1165 3 x 50-bit     */
    CNTT_NATIVE_BINARY32_PLAN52 = 8,  /* native_binary32::Plan52   src/native_binary32.rs:19,256-322 1 x 50-bit   */
    CNTT_NATIVE_BINARY64_PLAN52 = 9   /* native_binary64::Plan52   src/native_binary64.rs:29,449-521 2 x 50-bit   */
} cntt_native_kind_t;

typedef struct cntt_native cntt_native_t;

int cntt_native_plan_new(cntt_native_kind_t kind, size_t n, cntt_native_t **out);   /* PlanNN::try_new(n)  e.g. src/native64.rs:933-942 */
cntt_native_t *cntt_native_plan_clone(const cntt_native_t *plan);
void cntt_native_plan_free(cntt_native_t *plan);
size_t cntt_native_ntt_size(const cntt_native_t *plan);                             /* ntt_size()  src/native64.rs:946-948 */
int cntt_native_nprimes(const cntt_native_t *plan);
int cntt_native_word_bytes(const cntt_native_t *plan);                              /* 4, 8 or 16 */
int cntt_native_residue_bytes(const cntt_native_t *plan);                           /* 4 (Plan32) or 8 (Plan52) */
/* borrowed sub-plans: ntt_0() .. ntt_k()  src/native64.rs:950-969 ; NULL if the kind does not match */
const cntt_plan32_t *cntt_native_ntt32(const cntt_native_t *plan, int i);
const cntt_plan64_t *cntt_native_ntt64(const cntt_native_t *plan, int i);

/* fwd(value, mod_p0..)         src/native64.rs:971-999   : residues[i] <- NTT_i(value mod P_i) */
int cntt_native_fwd(const cntt_native_t *plan, const void *value, size_t len, void *const *residues);
/* fwd_binary(value, mod_p0..)  src/native_binary64.rs:372-389 (binary kinds only) */
int cntt_native_fwd_binary(const cntt_native_t *plan, const void *value, size_t len, void *const *residues);
/* inv(value, mod_p0..)         src/native64.rs:1001-1038 : residues are inverse-transformed IN PLACE, then CRT */
int cntt_native_inv(const cntt_native_t *plan, void *value, size_t len, void *const *residues);
/* negacyclic_polymul(prod, lhs, rhs)  src/native64.rs:1042-1069 ; CNTT_ELEN unless the three lengths are equal */
int cntt_native_negacyclic_polymul(const cntt_native_t *plan, void *prod, size_t prod_len, const void *lhs, size_t lhs_len, const void *rhs, size_t rhs_len);

int cntt_native_fwd_batch(const cntt_native_t *plan, const void *value, void *const *residues, size_t batch, cntt_mem_t where, void *stream);
int cntt_native_fwd_binary_batch(const cntt_native_t *plan, const void *value, void *const *residues, size_t batch, cntt_mem_t where, void *stream);
int cntt_native_inv_batch(const cntt_native_t *plan, void *value, void *const *residues, size_t batch, cntt_mem_t where, void *stream);
/* The device path of the Plan32 kinds is one kernel for 32 <= n <= 32768.  n <= 4096 (except native128) needs no
 * workspace (nor n = 8192 of native32 / native64 / native_binary64); the other n = 8192 kinds, n = 16384 / 32768 and native128 park
 * residue tiles in a per-plan, per-device workspace of a few tens of MiB (n = 32768: 64 ... 300 MiB, native128 the most),
 * independent of the batch; the Plan52 kinds (round 5) run the whole-product kernel of the Plan32 kind with the same words -- the
 * wrapping product does not depend on the primes it is computed with -- and fall back to the composed pipeline on a workspace of
 * 2 * nprimes * batch * n residues where that kind does not exist (n < 32) or when the testing switch "plan52_via32" is 0.  The workspace grows on demand (an allocation, and a device synchronisation when it
 * is replaced): reserve it ahead of a timed or captured region with cntt_native_reserve().  Calls on different streams
 * that share a plan are ordered against each other by the library wherever they share the workspace -- EXCEPT while a
 * stream is being captured into a hipGraph: a captured launch neither waits on nor records the workspace event, and it
 * bakes the workspace address into the graph.  Rules for captured use: reserve the largest batch BEFORE the capture and
 * never grow the workspace afterwards (a regrow frees the buffer a captured graph still points at); do not replay such a
 * graph concurrently with eager calls, or with other graphs, of the SAME plan on another stream (the persistent kernels
 * park tiles at workgroup-indexed offsets of the one workspace) -- use one plan (cntt_native_plan_clone) per concurrent
 * stream. */
int cntt_native_negacyclic_polymul_batch(const cntt_native_t *plan, void *prod, const void *lhs, const void *rhs, size_t batch, cntt_mem_t where, void *stream);
int cntt_native_reserve(const cntt_native_t *plan, size_t batch);

/* ===================================================================================== */
/* product::Plan  (src/product.rs:139-149): negacyclic NTT modulo a product of distinct    */
/* primes that fits u64 -- the interface tfhe-rs calls.                                    */
/*                                                                                       */
/* NTT-domain buffer of ONE polynomial, in u64 words (src/product.rs:261-270): the u32     */
/* residues of the primes below 2^32 (ascending, n u32 each, two per word), then the u64   */
/* residues of the other primes (n words each); cntt_product_ntt_domain_len() words.       */
/* _batch calls take `batch` polynomials: `standard` back to back (b at base + b*n), the   */
/* NTT domain PLANE-MAJOR: for each prime in that same order, the residues of all `batch`  */
/* polynomials back to back (plane of prime k holds batch*n residues).  batch == 1 is the  */
/* reference layout.                                                                       */
/* ===================================================================================== */
typedef struct cntt_product cntt_product_t;
typedef enum cntt_fwd_mode { CNTT_FWD_GENERIC = 0, CNTT_FWD_BOUNDED = 1 } cntt_fwd_mode_t;    /* FwdMode src/product.rs:124-129; Bounded(bound) */
typedef enum cntt_inv_mode { CNTT_INV_REPLACE = 0, CNTT_INV_ACCUMULATE = 1 } cntt_inv_mode_t; /* InvMode src/product.rs:131-136 */

/* Plan::try_new(polynomial_size, modulus, factors) src/product.rs:153-247.  CNTT_NONE when the reference returns
 * None: odd size, a zero or repeated factor, product of the factors (1s skipped) != modulus or overflowing u64,
 * or a factor for which prime32/prime64::Plan::try_new is None. */
int cntt_product_plan_new(size_t polynomial_size, uint64_t modulus, const uint64_t *factors, size_t nfactors, cntt_product_t **out);
cntt_product_t *cntt_product_plan_clone(const cntt_product_t *plan);   /* #[derive(Clone)] src/product.rs:138 */
void cntt_product_plan_free(cntt_product_t *plan);
size_t cntt_product_ntt_size(const cntt_product_t *plan);              /* src/product.rs:251-253 */
uint64_t cntt_product_modulus(const cntt_product_t *plan);             /* src/product.rs:257-259 */
size_t cntt_product_ntt_domain_len(const cntt_product_t *plan);        /* src/product.rs:268-270 */
int cntt_product_nprimes32(const cntt_product_t *plan);                /* plan_32.len() */
int cntt_product_nprimes64(const cntt_product_t *plan);                /* plan_64.len() */
uint64_t cntt_product_prime(const cntt_product_t *plan, int i);        /* i-th factor after sorting and dropping 1s */
const cntt_plan32_t *cntt_product_ntt32(const cntt_product_t *plan, int i); /* borrowed plan_32[i] */
const cntt_plan64_t *cntt_product_ntt64(const cntt_product_t *plan, int i); /* borrowed plan_64[i] */
/* private field modular_inverses (src/product.rs:207-229), len = k(k-1)/2, for parity tests */
int cntt_product_modular_inverses(const cntt_product_t *plan, uint64_t *out, size_t len);

/* Plan::fwd(ntt, standard, mode) src/product.rs:273-357; `bound` is read only for CNTT_FWD_BOUNDED */
int cntt_product_fwd(const cntt_product_t *plan, uint64_t *ntt, size_t ntt_len, const uint64_t *standard, size_t standard_len, cntt_fwd_mode_t mode, uint64_t bound);
/* Plan::inv(standard, ntt, mode) src/product.rs:360-879; like the reference it leaves the inverse-transformed residues in `ntt` */
int cntt_product_inv(const cntt_product_t *plan, uint64_t *standard, size_t standard_len, uint64_t *ntt, size_t ntt_len, cntt_inv_mode_t mode);
int cntt_product_mul_assign_normalize(const cntt_product_t *plan, uint64_t *lhs, size_t lhs_len, const uint64_t *rhs, size_t rhs_len); /* src/product.rs:885-913 */
int cntt_product_normalize(const cntt_product_t *plan, uint64_t *values, size_t len);                                                  /* src/product.rs:917-931 */
int cntt_product_mul_accumulate(const cntt_product_t *plan, uint64_t *acc, size_t acc_len, const uint64_t *lhs, size_t lhs_len, const uint64_t *rhs, size_t rhs_len); /* src/product.rs:935-966 */

int cntt_product_fwd_batch(const cntt_product_t *plan, uint64_t *ntt, const uint64_t *standard, size_t batch, cntt_fwd_mode_t mode, uint64_t bound, cntt_mem_t where, void *stream);
int cntt_product_inv_batch(const cntt_product_t *plan, uint64_t *standard, uint64_t *ntt, size_t batch, cntt_inv_mode_t mode, cntt_mem_t where, void *stream);
int cntt_product_mul_assign_normalize_batch(const cntt_product_t *plan, uint64_t *lhs, const uint64_t *rhs, size_t batch, cntt_mem_t where, void *stream);
int cntt_product_normalize_batch(const cntt_product_t *plan, uint64_t *values, size_t batch, cntt_mem_t where, void *stream);
int cntt_product_mul_accumulate_batch(const cntt_product_t *plan, uint64_t *acc, const uint64_t *lhs, const uint64_t *rhs, size_t batch, cntt_mem_t where, void *stream);
/* The external-product step of the reference's caller (tfhe-rs NTT backend) in one call:
 *     for j < nterms { plan.fwd(t_j, terms[b][j], fwd_mode); for o < nout { plan.mul_accumulate(acc_o, t_j, key[j][o]) } }
 *     for o { plan.inv(out[b][o], acc_o, inv_mode) }                    (acc_o starts at zero)
 * terms: batch x nterms polynomials of n words (standard domain, element b's terms back to back); key_ntt: nterms x nout
 * polynomials in the NTT domain in the plane-major batched layout above (i.e. what cntt_product_fwd_batch writes for a
 * batch of nterms*nout polynomials, key[j][o] at batch index j*nout + o); out: batch x nout polynomials of n words.
 * Same values as the separate calls.  Runs as: one residue split, one fused mul_accumulate chain per prime
 * (cntt_prime*_external_product_batch), one Garner recombination; residues live in a stream-ordered scratch buffer. */
int cntt_product_external_product_batch(const cntt_product_t *plan, uint64_t *out, const uint64_t *terms, const uint64_t *key_ntt, size_t nterms, size_t nout, size_t batch, cntt_fwd_mode_t fwd_mode, uint64_t bound, cntt_inv_mode_t inv_mode, cntt_mem_t where, void *stream);

/* ===================================================================================== */
/* utilities (not part of the reference API)                                              */
/* ===================================================================================== */
/* synthetic inputs: dst[i] = mulhi64(splitmix64(seed + i), bound); bound == 0 -> raw 64 bits.
 * u32 variant: ((splitmix64(seed + i) >> 32) * bound) >> 32.  Device memory only. */
int cntt_fill_uniform_u64(uint64_t *dst, size_t count, uint64_t bound, uint64_t seed, void *stream);
int cntt_fill_uniform_u32(uint32_t *dst, size_t count, uint32_t bound, uint64_t seed, void *stream);
#ifdef __cplusplus
}
#endif
#endif /* CNTT_H */
