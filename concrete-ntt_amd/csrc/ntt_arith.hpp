// Device-side modular arithmetic for the NTT / pointwise / CRT kernels (gfx950).
//
// Every public result of the reference is canonical (< p) -- asserted by its tests
// (src/prime64.rs:1234-1252, src/prime32.rs:1028-1045) -- so the device is free to pick its own
// internal ranges; what must match is the mathematics, the root of unity and the table layout.
// Three arithmetic classes, chosen per plan at creation from the modulus:
//   CLS_LAZY    p < 2^(B-2): Harvey lazy butterflies, values in [0,4p) (fwd) / [0,2p) (inv), as
//               src/prime64/less_than_62bit.rs:117-154,271-310 and src/prime32/less_than_30bit.rs
//   CLS_STRICT  p < 2^(B-1): the butterflies of src/prime64/less_than_63bit.rs:117-154,214-232 (src/prime32/less_than_31bit.rs)
//               operation for operation -- words below p, every sum and Shoup product brought back by ONE min(x, x - p).
//               (Rounds 1-3 kept this class lazy in [0, 2p) with carry-aware corrections: same results on canonical inputs,
//               but the reference's own mul_accumulate can hand its inverse transform a word in [p, 2^B - p) for these
//               primes -- DESIGN 4 -- and only the same instruction sequence returns the same words then.)
//   CLS_GENERIC any p (used for p >= 2^(B-1), incl. Solinas): canonical values, Montgomery
//               products against twiddles stored in Montgomery form.  The reference does exact
//               `%`-products there (src/prime64/generic_solinas.rs:42-128): same values.
//   CLS_FP      64-bit words, p < 2^50 (the reference's src/prime64/less_than_50bit.rs class, and every Plan52
//               native plan): the LDS-resident transforms keep each residue as an IEEE double holding an exact
//               integer representative |v| < 2^53 and multiply with v_fma_f64 -- on gfx950 a double FMA
//               issues at the rate of ONE 32-bit integer multiply, and an exact 50 x 53-bit modular product
//               is six of them instead of ten integer multiplies and their carry chains (see BflyFp).
//   CLS_PM64    64-bit words, p = 2^64 - c with c < 2^32: the Solinas prime 2^64 - 2^32 + 1 (src/prime64/
//               generic_solinas.rs:35-40,102-128) and every "largest prime below 2^64" of the reference's benches.
//               2^64 = c (mod p) folds a 128-bit product in two multiply-adds instead of a Montgomery reduction.
//   CLS_FP51    the same for 2^50 <= p < 2^51 (src/prime64/less_than_51bit.rs): 2^53 is only 4p there, so the range
//               reductions come more often.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace cntt {

enum : int { CLS_LAZY = 0, CLS_STRICT = 1, CLS_GENERIC = 2, CLS_FP = 3, CLS_FP51 = 4, CLS_PM64 = 5, CLS_FPW = 6 };

// One table entry: the twiddle and its Shoup companion floor(w * 2^B / p) (CLS_GENERIC: w holds
// w * 2^B mod p and ws is unused).  Interleaved so that one 16-byte (u64) / 8-byte (u32) load
// fetches both.
template <class T> struct alignas(2 * sizeof(T)) TwPair {
    T w, ws;
};

// Per-plan scalars handed to every kernel by value (they live in SGPRs).
template <class T> struct ModParams {
    T p;        // modulus
    T neg_p;    // 2^B - p
    T two_p;    // 2p (LAZY/STRICT; wraps and is unused for GENERIC)
    T neg_two_p; // 2^B - 2p
    T pinv_neg; // -p^-1 mod 2^B (Montgomery, GENERIC only)
    T n_inv, n_inv_shoup;  // N^-1 mod p and its Shoup companion (GENERIC: N^-1 * 2^B mod p, unused)
    T p_barrett;           // floor(2^(big_q + B - 1) / p)   (src/prime64.rs:754-756)
    T r2;                  // 2^(2B) mod p (GENERIC pointwise)
    T last_w, last_w_shoup; // inv_twid[1] * N^-1 mod p and its Shoup companion (GENERIC: * 2^(2B), unused): the
                           // fused product kernel normalises inside the last inverse stage (Bfly::inv_norm)
    // CLS_FP (64-bit words, p < 2^50): bit patterns of doubles -- p, 1/p, and the plan constants in centred
    // form c in (-p/2, p/2] with their quotient companions c/p
    // (also CLS_FPW: 32-bit words whose transforms run on doubles)
    uint64_t fp_p, fp_pinv, fp_n_inv, fp_n_inv_q, fp_last_w, fp_last_w_q;
    uint32_t big_q;        // floor(log2 p) + 1
    uint32_t cls;          // integer arithmetic class (pointwise kernels, global stages, every non-FP transform)
    uint32_t fp;           // CLS_FP / CLS_FP51 / CLS_FPW: class of the LDS-resident transforms of this plan (0: cls)
    // CLS_PM64 (p = 2^64 - c, c < 2^32): c, and the plan constants as plain residues
    uint32_t pm_c;         // 0: not such a modulus
    T pm_n_inv, pm_last_w;
    // lazy class, fused product kernels (round 4): N^-1 2^B and inv_twid[1] N^-1 2^B mod p with their Shoup companions -- the pointwise
    // product between the transforms is a Montgomery product there (mul_fused) and leaves a factor 2^-B for the last inverse stage
    T mont_n_inv, mont_n_inv_shoup, mont_last_w, mont_last_w_shoup;
    T mont_r, mont_r_shoup;   // 2^B mod p and its Shoup companion: the fused mul_accumulate chains undo their products' 2^-B with it
    // set per LAUNCH by the host (0 in the plan): the batch of this call is larger than the 256 MiB Infinity Cache and passes through the
    // chip once -- the stand-alone transform kernels then put the non-temporal hint on their tile loads and stores (ntt_kernel.hpp)
    uint32_t stream;
};

// ---------------------------------------------------------------------------------------------
// wide multiplies
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t mulhi(uint32_t a, uint32_t b) { return __umulhi(a, b); }

__device__ __forceinline__ uint64_t mulhi(uint64_t a, uint64_t b) {
    // 64x64 -> high 64 from four 32x32 products.  The two cross products are summed through a 96-bit
    // carry chain (v_add_co / v_addc / v_addc) whose upper 64 bits feed the last v_mad_u64_u32 directly:
    // 1 v_mul_hi_u32 + 3 v_mad_u64_u32 + 3 adds, no zero-extension moves.
    const uint32_t a0 = (uint32_t)a, a1 = (uint32_t)(a >> 32);
    const uint32_t b0 = (uint32_t)b, b1 = (uint32_t)(b >> 32);
    const uint64_t t = __umulhi(a0, b0);
    const uint64_t u = (uint64_t)a1 * b0 + t;  // cannot overflow: (2^32-1)^2 + 2^32-1 < 2^64
    const uint64_t v = (uint64_t)a0 * b1;
    const uint64_t z = (uint64_t)(((unsigned __int128)u + v) >> 32);
    return (uint64_t)a1 * b1 + z;              // cannot overflow: the true high word is < 2^64
}

// low B bits of a*b + c*d (wrapping)
__device__ __forceinline__ uint32_t mullo2(uint32_t a, uint32_t b, uint32_t c, uint32_t d) { return a * b + c * d; }

// ---------------------------------------------------------------------------------------------
// 64-bit Shoup product with an optional addend:
//     shoup_core(y, w, ws, neg_p, x) = x + y*w + floor(y*ws / 2^64) * neg_p      (mod 2^64)
// i.e. x + (y*w mod p) with the product left in [0, 2p) (any y < 2^64, w < p < 2^63, ws = floor(w 2^64 / p)),
// in thirteen VALU instructions (ten multiplies).  The middle eight are one inline-asm block because two things
// cannot be said in C++:
//   * the carry of the cross-product sum y0*s1 + (y1*s0 + hi(y0*s0)) comes out of v_mad_u64_u32's own carry-out
//     (v_mov + v_cndmask rebuild bits 32..95 of the sum instead of a three-instruction 96-bit add chain);
//   * the four cross products of the low halves only matter modulo 2^32: they are accumulated in the LOW word of
//     one v_mad_u64_u32 chain whose upper word is scratch (hipcc would narrow them to v_mul_lo_u32 + v_add).
// The addend x rides in src2 of the low product y0*w0, so "x + t" costs nothing.  The quotient is handed back in
// the fixed pair v[2:3]: VGPR tuples must be 64-bit aligned on gfx950 and the halves of a tuple cannot be named
// through an asm operand, while the block has to address both halves (v_mov / v_cndmask, 32-bit multiplicands).
// gfx950 needs two wait states between a VALU write of an SGPR and a VALU read of it: the v_cndmask sits three
// instructions behind the v_mad that produces the carry.
// UNI: w and ws are wave-uniform (SGPRs: first-pass twiddles, plan constants) -- one scalar source per
// instruction, as the constant bus of this ISA allows.
// ---------------------------------------------------------------------------------------------
#define CNTT_SHOUP_BODY                                          \
    "v_mad_u64_u32 v[2:3], %[c], %[y0], %[s1], %[u]\n\t"        \
    "v_mad_u64_u32 %[h], vcc, %[y0], %[w1], 0\n\t"              \
    "v_mad_u64_u32 %[h], vcc, %[y1], %[w0], %[h]\n\t"           \
    "v_mov_b32 v2, v3\n\t"                                      \
    "v_cndmask_b32_e64 v3, 0, 1, %[c]\n\t"                      \
    "v_mad_u64_u32 v[2:3], vcc, %[y1], %[s1], v[2:3]\n\t"       \
    "v_mad_u64_u32 %[h], vcc, v2, %[n1], %[h]\n\t"              \
    "v_mad_u64_u32 %[h], vcc, v3, %[n0], %[h]"

template <bool UNI, bool ADD>
__device__ __forceinline__ uint64_t shoup_core(uint64_t y, uint64_t w, uint64_t ws, uint64_t neg_p, uint64_t x) {
    const uint32_t y0 = (uint32_t)y, y1 = (uint32_t)(y >> 32);
    const uint32_t s0 = (uint32_t)ws, s1 = (uint32_t)(ws >> 32);
    const uint32_t w0 = (uint32_t)w, w1 = (uint32_t)(w >> 32);
    const uint32_t n0 = (uint32_t)neg_p, n1 = (uint32_t)(neg_p >> 32);
    const uint64_t t = __umulhi(y0, s0);
    const uint64_t u = (uint64_t)y1 * s0 + t;  // cannot overflow
    uint64_t q, h, carry;
    if constexpr (UNI)
        asm(CNTT_SHOUP_BODY : "=&{v[2:3]}"(q), [h] "=&v"(h), [c] "=&s"(carry)
            : [y0] "v"(y0), [y1] "v"(y1), [s1] "s"(s1), [w0] "s"(w0), [w1] "s"(w1), [n0] "s"(n0), [n1] "s"(n1), [u] "v"(u)
            : "vcc");
    else
        asm(CNTT_SHOUP_BODY : "=&{v[2:3]}"(q), [h] "=&v"(h), [c] "=&s"(carry)
            : [y0] "v"(y0), [y1] "v"(y1), [s1] "v"(s1), [w0] "v"(w0), [w1] "v"(w1), [n0] "s"(n0), [n1] "s"(n1), [u] "v"(u)
            : "vcc");
    uint64_t acc = (uint64_t)y0 * w0;
    if constexpr (ADD) acc += x;                      // one v_mad_u64_u32
    uint32_t hi = (uint32_t)(acc >> 32) + (uint32_t)h;  // v_add_u32: the cross sum only reaches the upper word
    asm("" : "+v"(hi));                               // (opaque, or hipcc turns it into a move and a 64-bit add)
    acc = ((uint64_t)hi << 32) | (uint32_t)acc;
    return (uint64_t)(uint32_t)q * n0 + acc;          // wraps modulo 2^64, as wanted
}

template <class T> __device__ __forceinline__ T umin(T a, T b) { return a < b ? a : b; }

// x in [0, 2m) -> [0, m)
template <class T> __device__ __forceinline__ T csub(T x, T m) { return umin<T>(x, x - m); }

// ---------------------------------------------------------------------------------------------
// 64-bit conditional subtraction of a value known to lie in [0, 2m), m < 2^63, WITHOUT the condition-code register (round 5): x - m then
// lies in [-m, m), so its sign bit IS the borrow; v_ashrrev_i32 spreads it to a lane mask and two v_bitop3_b32 pick the result -- four
// instructions at the plain rate, no scalar write, no wait state.  (The natural form -- v_cmp_lt_u64 + 2 v_cndmask_b32 -- writes VCC from
// the vector ALU and reads it back: two wait states per read on gfx950.)  Measured A/B, profiles/r05_csub_mask_ab.txt: the same VALU count
// (2010 against 2013 per thread and polynomial at N = 1024), half the padded wait states (188 against 374), the stages on registers 3 %
// faster in the inverse direction, the whole kernels unchanged -- with eight independent butterflies per stage the scheduler had already
// hidden the wait states.  Kept because it is never slower and frees VCC for the carry chains around it.
// ---------------------------------------------------------------------------------------------
// all-ones if bit 31 of w is set, else zero (v_ashrrev_i32)
__device__ __forceinline__ uint32_t top_bit_mask(uint32_t w) { return (uint32_t)((int32_t)w >> 31); }
// m ? a : b, bit by bit: two v_bitop3_b32 (truth table 0xca = (m & a) | (~m & b)).  Written as the builtin: from the plain expression hipcc
// recognises a select on a sign and goes back to v_cmp_gt_i64 + v_cndmask; behind an opaque asm it pads a wait state after the mask.
__device__ __forceinline__ uint64_t mask_select(uint32_t m, uint64_t a, uint64_t b) {
    const uint32_t lo = __builtin_amdgcn_bitop3_b32(m, (uint32_t)a, (uint32_t)b, 0xca);
    const uint32_t hi = __builtin_amdgcn_bitop3_b32(m, (uint32_t)(a >> 32), (uint32_t)(b >> 32), 0xca);
    return ((uint64_t)hi << 32) | lo;
}
__device__ __forceinline__ uint64_t csub_sign(uint64_t x, uint64_t neg_m) {
    const uint64_t d = x + neg_m;   // v_lshl_add_u64
    return mask_select(top_bit_mask((uint32_t)(d >> 32)), x, d);
}

// min(x, x - p) for the strict class's butterflies, ANY word x (the class mirrors the reference operation for operation, so the sign of
// the difference is not enough): the borrow of the subtraction is the select condition (four instructions and two s_nop for 64 bits;
// hipcc recomputes the condition with a 64-bit compare: five and the same two s_nop), p in scalar registers.
// (Round 5 tried the VCC-free form here as well -- borrow rebuilt from the top words by v_bitop3_b32 0x8e, mask, two selects: five plain
// instructions.  VALU count per thread 2363 -> 2562, padded wait states 891 -> 205, and the kernels 3 % SLOWER (316 -> 327 us forward,
// 316 -> 320 us inverse per 65 536 transforms, profiles/r05_csub_mask_ab.txt): the extra instruction costs more than the hidden waits.)
template <class T> __device__ __forceinline__ T csub_p(T x, T p) {
    if constexpr (sizeof(T) == 8) {
        const uint32_t x0 = (uint32_t)x, x1 = (uint32_t)(x >> 32), p0 = (uint32_t)p, p1 = (uint32_t)(p >> 32);
        uint32_t r0, r1;
        // (the high half of p in a VGPR: the carry-in already takes v_subb's one scalar operand.  gfx950 has no interlock between a VALU
        // write of VCC and a VALU read of it -- two wait states, which hipcc pads in its own code and nobody pads inside an asm string)
        asm("v_subrev_co_u32 %0, vcc, %4, %2\n\t"
            "s_nop 1\n\t"
            "v_subb_co_u32 %1, vcc, %3, %5, vcc\n\t"
            "s_nop 1\n\t"
            "v_cndmask_b32 %0, %0, %2, vcc\n\t"
            "v_cndmask_b32 %1, %1, %3, vcc"
            : "=&v"(r0), "=&v"(r1) : "v"(x0), "v"(x1), "s"(p0), "v"(p1) : "vcc");
        return ((uint64_t)r1 << 32) | r0;
    } else {
        return umin<T>(x, x - p);
    }
}

// x in [0, 4p) -> [0, 2p)   (p < 2^(B-2))
template <class T> __device__ __forceinline__ T csub_two_p(T x, T two_p, T neg_two_p) {
    if constexpr (sizeof(T) == 8) return csub_sign(x, neg_two_p);
    else return umin<T>(x, x - two_p);
}
// x in [0, 2p) -> [0, p)   (p < 2^(B-1))
template <class T> __device__ __forceinline__ T csub_one_p(T x, T p, T neg_p) {
    if constexpr (sizeof(T) == 8) return csub_sign(x, neg_p);
    else return umin<T>(x, x - p);
}

// Shoup product: y * w - floor(y * ws / 2^B) * p, in [0, 2p) for any y < 2^B  (needs p < 2^(B-1))
template <class T, bool UNI = false> __device__ __forceinline__ T shoup_mul(T y, T w, T ws, T neg_p) {
    if constexpr (sizeof(T) == 8) {
        return shoup_core<UNI, false>(y, w, ws, neg_p, 0);
    } else {
        const T q = mulhi(y, ws);
        return mullo2(y, w, q, neg_p);
    }
}
// x + shoup_mul(y, w, ws): for 64 bits the sum is free (the addend of the product's own multiply-add chain)
template <class T, bool UNI = false> __device__ __forceinline__ T shoup_mad(T y, T w, T ws, T neg_p, T x) {
    if constexpr (sizeof(T) == 8) {
        return shoup_core<UNI, true>(y, w, ws, neg_p, x);
    } else {
        return x + shoup_mul<T>(y, w, ws, neg_p);
    }
}

// ---------------------------------------------------------------------------------------------
// Montgomery (GENERIC): works for every odd p < 2^B, values canonical
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t mont_mul(uint32_t a, uint32_t b, uint32_t p, uint32_t pinv_neg) {
    const uint64_t t = (uint64_t)a * b;
    const uint32_t m = (uint32_t)t * pinv_neg;
    const uint64_t mp = (uint64_t)m * p;
    // (t + mp) / 2^32, 65-bit safe: low words cancel, carry = (low(t) != 0)
    const uint64_t u = (t >> 32) + (mp >> 32) + ((uint32_t)t != 0u);  // < 2p <= 2^33
    return (uint32_t)(u >= p ? u - p : u);
}
__device__ __forceinline__ uint64_t mont_mul(uint64_t a, uint64_t b, uint64_t p, uint64_t pinv_neg) {
    const uint64_t t_lo = a * b, t_hi = mulhi(a, b);
    const uint64_t m = t_lo * pinv_neg;
    const uint64_t mp_hi = mulhi(m, p);
    const uint64_t s = t_hi + mp_hi;
    const bool c1 = s < t_hi;
    const uint64_t u = s + (t_lo != 0u);
    const bool c2 = u < s;
    return (c1 || c2 || u >= p) ? u - p : u;
}
template <class T> __device__ __forceinline__ T add_mod(T a, T b, T p) {  // canonical in/out, any p < 2^B
    const T neg_b = p - b;
    return a >= neg_b ? a - neg_b : a + b;
}
template <class T> __device__ __forceinline__ T sub_mod(T a, T b, T p) {
    return a >= b ? a - b : a + (p - b);
}

// ---------------------------------------------------------------------------------------------
// butterflies
// ---------------------------------------------------------------------------------------------
template <class T, int CLS> struct Bfly {
    // forward (Cooley-Tukey): (x, y) <- (x + w y, x - w y).  UNI: w, ws are wave-uniform (scalar registers)
    // FIRST: the first stage of a whole transform -- x is an input coefficient, canonical by the API's contract (< p, as
    // the reference's tests feed it: SURVEY 8(a5)), so the lazy class's conditional subtraction of 2p is a no-op and is
    // left out (x < 2p is all the butterfly needs)
    template <bool UNI = false, bool FIRST = false>
    static __device__ __forceinline__ void fwd(T &x, T &y, T w, T ws, const ModParams<T> &P) {
        if constexpr (CLS == CLS_LAZY) {
            if constexpr (!FIRST) x = csub_two_p<T>(x, P.two_p, P.neg_two_p);
            if constexpr (sizeof(T) == 8) {
                const T xn = shoup_mad<T, UNI>(y, w, ws, P.neg_p, x);  // x + t, t in [0, 2p)
                y = ((x << 1) + P.two_p) - xn;                         // x - t + 2p
                x = xn;
            } else {
                const T t = shoup_mul<T>(y, w, ws, P.neg_p);
                y = (x + P.two_p) - t;
                x = x + t;
            }
        } else if constexpr (CLS == CLS_STRICT) {
            // fwd_butterfly_scalar (less_than_63bit.rs:117-133): z0 = min(z0, z0 - p); t = min(t, t - p); (z0 + t, z0 - t + p), both in [0, 2p)
            const T z0 = csub_p<T>(x, P.p);
            const T t = csub_p<T>(shoup_mul<T, UNI>(y, w, ws, P.neg_p), P.p);
            x = z0 + t;
            y = (z0 - t) + P.p;
        } else {
            const T t = mont_mul(y, w, P.p, P.pinv_neg);
            const T x0 = x;
            x = add_mod<T>(x0, t, P.p);
            y = sub_mod<T>(x0, t, P.p);
        }
    }
    // inverse (Gentleman-Sande): (x, y) <- (x + y, (x - y) w)
    // Every stage reduces its sum, the first one included (less_than_62bit.rs:282-283): the reference's own mul_accumulate can leave a
    // word in [p, 2p) for lazy primes just above a power of two (its Barrett estimate reaches 2p there), and fwd -> mul_accumulate ->
    // inv must still come out right.  (Round 4 skipped the first stage's reduction for "canonical" memory words: ADVICE round 4.)
    template <bool UNI = false, bool FIRST = false>
    static __device__ __forceinline__ void inv(T &x, T &y, T w, T ws, const ModParams<T> &P) {
        if constexpr (CLS == CLS_LAZY) {
            const T d = (x + P.two_p) - y;
            x = csub_two_p<T>(x + y, P.two_p, P.neg_two_p);
            y = shoup_mul<T, UNI>(d, w, ws, P.neg_p);
        } else if constexpr (CLS == CLS_STRICT) {
            // inv_butterfly_scalar (less_than_63bit.rs:214-232): words below p in, words below p out; wrapping sums as there
            const T t = (x - y) + P.p;
            x = csub_p<T>(x + y, P.p);
            y = csub_p<T>(shoup_mul<T, UNI>(t, w, ws, P.neg_p), P.p);
        } else {
            const T x0 = x;
            x = add_mod<T>(x0, y, P.p);
            y = mont_mul(sub_mod<T>(x0, y, P.p), w, P.p, P.pinv_neg);
        }
    }
    // last inverse stage with the 1/N normalisation folded in: (x, y) <- ((x + y) / N, (x - y) w / N).
    // w = inv_twid[1] for every butterfly of that stage, so P.last_w = w / N replaces the twiddle and only the
    // sum branch pays one extra product (N/2 per polynomial instead of the N of a separate normalize pass).
    static __device__ __forceinline__ void inv_norm(T &x, T &y, const ModParams<T> &P) {
        inv<true>(x, y, P.last_w, P.last_w_shoup, P);
        if constexpr (CLS == CLS_LAZY) {
            x = shoup_mul<T, true>(x, P.n_inv, P.n_inv_shoup, P.neg_p);  // [0, 2p)
        } else if constexpr (CLS == CLS_STRICT) {
            x = csub_one_p<T>(shoup_mul<T, true>(x, P.n_inv, P.n_inv_shoup, P.neg_p), P.p, P.neg_p);  // (finish_inv is the identity for this class)
        } else {
            x = mont_mul(x, P.n_inv, P.p, P.pinv_neg);  // n_inv field = N^-1 R^2: x / R * (N^-1 R^2) / R ... see mul_for_inv
        }
    }
    static constexpr bool IS_FP = false;
    static constexpr bool FUSED_LAZY = false;  // the fused kernels hand canonical values to the pointwise product
    // a word as loaded from memory -> the class's register form (identity for the integer classes)
    static __device__ __forceinline__ T load_fix(T v, const ModParams<T> &) { return v; }
    static __device__ __forceinline__ T reduce(T v, const ModParams<T> &) { return v; }
    // an accumulator of the fused chains -> what the inverse transform's first stage accepts
    static __device__ __forceinline__ T pre_inverse(T v, const ModParams<T> &) { return v; }
    // bring a value left by the last stage into [0, p)
    static __device__ __forceinline__ T finish_fwd(T v, const ModParams<T> &P) {
        if constexpr (CLS == CLS_LAZY) return csub_one_p<T>(csub_two_p<T>(v, P.two_p, P.neg_two_p), P.p, P.neg_p);
        if constexpr (CLS == CLS_STRICT) return csub_p<T>(v, P.p);
        return v;
    }
    static __device__ __forceinline__ T finish_inv(T v, const ModParams<T> &P) {
        if constexpr (CLS == CLS_LAZY) return csub_one_p<T>(v, P.p, P.neg_p);
        return v;   // (strict class: the reference's inverse butterflies leave their outputs as they are -- so does this one)
    }
};

// ---------------------------------------------------------------------------------------------
// 32-bit lazy class in 64-bit "boxes" (registers only): a coefficient is the LOW word of a 64-bit VGPR pair whose high
// word is don't-care.  v_mad_u64_u32 takes such a pair as its addend, so the forward butterfly's
//     x' = x + y w + q (-p)     (mod 2^32, q = hi32(y ws))
// is two multiply-adds instead of two v_mul_lo_u32, a subtraction and an addition: 7 instructions per butterfly instead
// of 8.5 (profiles/r02_butterfly_census.txt).  Every native / native_binary Plan32 product is 9..30 such transforms.
// The boxes exist inside a pass's register stages only; LDS and HBM hold plain 32-bit words.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ uint64_t box32(uint32_t v) {
    uint32_t junk;
    junk = __builtin_nondeterministic_value(junk);  // no instruction: the high word is whatever the register holds
    return ((uint64_t)junk << 32) | v;
}
template <bool UNI> __device__ __forceinline__ uint64_t mad_box(uint32_t a, uint32_t b, uint64_t c) {  // low word: a b + c
    uint64_t r;
    if constexpr (UNI) asm("v_mad_u64_u32 %0, vcc, %1, %2, %3" : "=v"(r) : "v"(a), "s"(b), "v"(c) : "vcc");
    else asm("v_mad_u64_u32 %0, vcc, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c) : "vcc");
    return r;
}
template <bool UNI> __device__ __forceinline__ uint64_t mul_box(uint32_t a, uint32_t b) {  // low word: a b
    uint64_t r;
    if constexpr (UNI) asm("v_mad_u64_u32 %0, vcc, %1, %2, 0" : "=v"(r) : "v"(a), "s"(b) : "vcc");
    else asm("v_mad_u64_u32 %0, vcc, %1, %2, 0" : "=v"(r) : "v"(a), "v"(b) : "vcc");
    return r;
}
template <int CLS> struct BoxOps {
    template <class T> struct USE { static constexpr bool value = sizeof(T) == 4 && CLS == CLS_LAZY; };
    using P32 = ModParams<uint32_t>;
    static __device__ __forceinline__ uint64_t box(uint32_t v, const P32 &) { return box32(v); }
    static __device__ __forceinline__ uint32_t unbox(uint64_t c, const P32 &) { return (uint32_t)c; }
    // (x, y) <- (x + w y, x - w y), values in [0, 4p)
    template <bool UNI, bool FIRST = false> static __device__ __forceinline__ void fwd(uint64_t &X, uint64_t &Y, uint32_t w, uint32_t ws, const P32 &P) {
        const uint32_t x = FIRST ? (uint32_t)X : umin<uint32_t>((uint32_t)X, (uint32_t)X - P.two_p);   // (FIRST: see Bfly::fwd)
        const uint32_t y = (uint32_t)Y;
        const uint32_t q = __umulhi(y, ws);
        const uint64_t a = mad_box<UNI>(y, w, box32(x));
        X = mad_box<true>(q, P.neg_p, a);                       // x + t, t = y w - q p in [0, 2p)
        Y = box32((x << 1) + P.two_p - (uint32_t)X);            // x - t + 2p
    }
    // (x, y) <- (x + y, (x - y) w), values in [0, 2p)
    template <bool UNI, bool FIRST = false> static __device__ __forceinline__ void inv(uint64_t &X, uint64_t &Y, uint32_t w, uint32_t ws, const P32 &P) {
        const uint32_t x = (uint32_t)X, y = (uint32_t)Y;
        const uint32_t d = (x + P.two_p) - y;
        const uint32_t s = x + y;
        X = box32(umin<uint32_t>(s, s - P.two_p));   // (every stage, the first included: see Bfly::inv)
        const uint32_t q = __umulhi(d, ws);
        Y = mad_box<true>(q, P.neg_p, mul_box<UNI>(d, w));
    }
    static __device__ __forceinline__ void inv_norm(uint64_t &X, uint64_t &Y, const P32 &P) {
        inv<true>(X, Y, P.last_w, P.last_w_shoup, P);
        const uint32_t x = (uint32_t)X;
        const uint32_t q = __umulhi(x, P.n_inv_shoup);
        X = mad_box<true>(q, P.neg_p, mul_box<true>(x, P.n_inv));   // [0, 2p)
    }
};

// ---------------------------------------------------------------------------------------------
// CLS_FP / CLS_FP51: residues as exact integers in doubles (64-bit words, p < 2^50 / p < 2^51)
//
// A register holds the bit pattern of a double v, an exact integer with v = residue (mod p) and |v| < 2^53.
// Product by a table constant c (centred, |c| <= p/2) with its companion cq = fl(c / p):
//     h = fl(y c)   l = fma(y, c, -h) (exact error term)   q = rint(fl(y cq))   r = fma(-q, p, h) (exact)   t = r + l
// y c - q p = t exactly, and |q - y c / p| <= 1/2 + |y| 2^-53, so |t| <= (1/2 + |y| 2^-53) p; h - q p = t - l is an
// integer below 2^53 because |l| <= ulp(h)/2 <= |y| p 2^-54.  Every sum x +- t is exact while it stays below 2^53,
// which the kernels guarantee by a range reduction v - p rint(v / p) (three instructions) every few stages.
// With H = 2^53 / p (> 8 for CLS_FP, > 4 for CLS_FP51) and bounds in units of p, B' = B + 1/2 + B/H per stage:
//   forward  (x, y) <- (x + t, x - t)
//     H = 8: from 1 (canonical input) 1.63, 2.33, 3.12, 4.01, 5.01; from 1/2 (after a reduction) 1.06 ... 4.11
//            -> every value is reduced after every FIFTH stage;
//     H = 4: from 1: 1.75, 2.69, 3.86; from 1/2: 1.13, 1.91, 2.88 -> after every THIRD stage;
//   inverse  (x, y) <- (x + y, (x - y) w): sums double, products come back below p
//     H = 8: inputs <= 1 give 2, then 4 -> the sums are reduced after every SECOND stage (|x - y| <= 4 always);
//     H = 4: the sums are reduced after EVERY stage (inputs <= 1, |x + y| <= 2, |x - y| <= 2, products <= 1).
// Twiddles come from a table of (c, c/p) doubles built by the host (csrc/host.hip); the public values are the same
// canonical integers as in every other class: the load turns a canonical word into a double (two instructions) and
// the store reduces, lifts negatives by p and extracts the integer (nine).
// ---------------------------------------------------------------------------------------------
struct Fp {
    static __device__ __forceinline__ double d(uint64_t v) { return __longlong_as_double((long long)v); }
    static __device__ __forceinline__ uint64_t u(double x) { return (uint64_t)__double_as_longlong(x); }
    // canonical word (< 2^52) -> double: 2^52 + v is the double whose mantissa field is v
    static __device__ __forceinline__ double from_word(uint64_t v) {
        return __dadd_rn(d(v | 0x4330000000000000ull), -4503599627370496.0);
    }
    // v - p rint(v / p): |result| <= (1/2 + 2^-49) p for |v| < 2^53
    static __device__ __forceinline__ double reduce(double v, double p, double pinv) {
        const double q = __builtin_rint(__dmul_rn(v, pinv));
        return __fma_rn(-q, p, v);
    }
    // y * c mod p for a table constant (c, cq = c / p)
    static __device__ __forceinline__ double mul_const(double y, double c, double cq, double p) {
        const double h = __dmul_rn(y, c);
        const double q = __builtin_rint(__dmul_rn(y, cq));
        const double l = __fma_rn(y, c, -h);
        const double r = __fma_rn(-q, p, h);
        return __dadd_rn(r, l);
    }
    // a * b mod p with the quotient from the rounded product (two data values, |a| <= p/2, 0 <= b < p < 2^51: |result| <=
    // 0.875 p; CLS_FPW: any |a b / p| < 2^47)
    static __device__ __forceinline__ double mul_data(double a, double b, double p, double pinv) {
        const double h = __dmul_rn(a, b);
        const double q = __builtin_rint(__dmul_rn(h, pinv));
        const double l = __fma_rn(a, b, -h);
        const double r = __fma_rn(-q, p, h);
        return __dadd_rn(r, l);
    }
    // |v| < 2^53 -> canonical word in [0, p)
    static __device__ __forceinline__ uint64_t to_word(double v, double p, double pinv) {
        double r = reduce(v, p, pinv);                    // |r| <= p/2 (+1)
        r = r < 0.0 ? __dadd_rn(r, p) : r;                // [0, p)
        return u(__dadd_rn(r, 4503599627370496.0)) & 0x000fffffffffffffull;  // mantissa field of 2^52 + r
    }
};

// HEAD = floor(2^53 / 2^bits(p)): 8 for p < 2^50, 4 for p < 2^51
template <class T, int HEAD> struct BflyFp {
    static_assert(sizeof(T) == 8, "the double-precision classes are 64-bit classes");
    static constexpr bool IS_FP = true;
    static constexpr bool FUSED_LAZY = true;   // mul_for_inv / mul_acc_cls take the lazy doubles
    static constexpr int FWD_REDUCE_EVERY = HEAD >= 8 ? 5 : 3;   // forward: every value, after this many stages
    static constexpr int INV_REDUCE_EVERY = HEAD >= 8 ? 2 : 1;   // inverse: the sums, after this many stages
    static constexpr int ACC_REDUCE_EVERY = HEAD >= 8 ? 8 : 2;   // mul_accumulate chains: products (<= 0.875 p) per reduction
    static __device__ __forceinline__ T load_fix(T v, const ModParams<T> &) { return Fp::u(Fp::from_word(v)); }
    static __device__ __forceinline__ T reduce(T v, const ModParams<T> &P) {
        return Fp::u(Fp::reduce(Fp::d(v), Fp::d(P.fp_p), Fp::d(P.fp_pinv)));
    }
    static __device__ __forceinline__ T pre_inverse(T v, const ModParams<T> &P) { return reduce(v, P); }  // |v| <= p
    template <bool UNI = false, bool FIRST = false>
    static __device__ __forceinline__ void fwd(T &x, T &y, T w, T ws, const ModParams<T> &P) {
        const double t = Fp::mul_const(Fp::d(y), Fp::d(w), Fp::d(ws), Fp::d(P.fp_p));
        const double xd = Fp::d(x);
        x = Fp::u(__dadd_rn(xd, t));
        y = Fp::u(__dadd_rn(xd, -t));
    }
    template <bool UNI = false, bool FIRST = false>
    static __device__ __forceinline__ void inv(T &x, T &y, T w, T ws, const ModParams<T> &P) {
        const double xd = Fp::d(x), yd = Fp::d(y);
        x = Fp::u(__dadd_rn(xd, yd));
        y = Fp::u(Fp::mul_const(__dadd_rn(xd, -yd), Fp::d(w), Fp::d(ws), Fp::d(P.fp_p)));
    }
    static __device__ __forceinline__ void inv_norm(T &x, T &y, const ModParams<T> &P) {
        const double xd = Fp::d(x), yd = Fp::d(y), p = Fp::d(P.fp_p);
        x = Fp::u(Fp::mul_const(__dadd_rn(xd, yd), Fp::d(P.fp_n_inv), Fp::d(P.fp_n_inv_q), p));
        y = Fp::u(Fp::mul_const(__dadd_rn(xd, -yd), Fp::d(P.fp_last_w), Fp::d(P.fp_last_w_q), p));
    }
    static __device__ __forceinline__ T finish_fwd(T v, const ModParams<T> &P) {
        return Fp::to_word(Fp::d(v), Fp::d(P.fp_p), Fp::d(P.fp_pinv));
    }
    static __device__ __forceinline__ T finish_inv(T v, const ModParams<T> &P) { return finish_fwd(v, P); }
};
template <class T> struct Bfly<T, CLS_FP> : BflyFp<T, 8> {};
template <class T> struct Bfly<T, CLS_FP51> : BflyFp<T, 4> {};
__host__ __device__ constexpr bool is_fp_class(int cls) { return cls == CLS_FP || cls == CLS_FP51; }

// ---------------------------------------------------------------------------------------------
// CLS_FPW: 32-bit WORDS whose transforms run on doubles (every odd p < 2^32; used for p >= 2^31, the moduli that have no
// lazy headroom in 32 bits and would otherwise take the Montgomery class at 20.6 instructions per butterfly).
//
// Inside a pass's register stages a coefficient is a double in a 64-bit "box" (as for the lazy 32-bit class, BoxOps):
// an exact integer v = residue (mod p).  Memory holds canonical words; LDS between the passes holds the CENTRED residue
// (|v| <= (p + 1) / 2 < 2^31) as an int32 bit pattern, so a pass boundary costs v_cvt_f64_i32 on the way in and
// reduce + v_cvt_i32_f64 on the way out.  A twiddle-table entry (8 bytes, like every TwPair<uint32_t>) IS the centred
// twiddle as a double (csrc/host.hip): no conversion, and no quotient companion -- the quotient comes from the product,
//     h = fl(y c)   q = rint(fl(h / p))   l = fma(y, c, -h)   r = fma(-q, p, h)   t = r + l        (Fp::mul_data)
// eight instructions per butterfly with the two sums.
// With p < 2^32 the doubles have H = 2^53 / p > 2^21 units of p of headroom, so NO range reduction is needed inside a
// transform: forward bounds grow by 1/2 + B / H per stage (<= 9 p after 16 stages), inverse sums double per stage
// (<= 2^15 p for the largest LDS-resident size) and the quotient estimate's error stays below 2^-5
// (|y c / p| <= 2^15 * 2^31 at 3 * 2^-53 relative), i.e. |t| <= (1/2 + 2^-5) p; h - q p = t - l is an integer far below
// 2^53, so r is exact.
// ---------------------------------------------------------------------------------------------
template <> struct BoxOps<CLS_FPW> {
    template <class T> struct USE { static constexpr bool value = sizeof(T) == 4; };
    using P32 = ModParams<uint32_t>;
    static __device__ __forceinline__ double tw(uint32_t lo, uint32_t hi) { return Fp::d(((uint64_t)hi << 32) | lo); }
    static __device__ __forceinline__ uint64_t box(uint32_t v, const P32 &) { return Fp::u((double)(int32_t)v); }  // v_cvt_f64_i32
    static __device__ __forceinline__ uint32_t unbox(uint64_t c, const P32 &P) {
        return (uint32_t)(int32_t)Fp::reduce(Fp::d(c), Fp::d(P.fp_p), Fp::d(P.fp_pinv));   // |.| <= (p + 1) / 2 < 2^31
    }
    template <bool UNI, bool FIRST = false> static __device__ __forceinline__ void fwd(uint64_t &X, uint64_t &Y, uint32_t w, uint32_t ws, const P32 &P) {
        const double x = Fp::d(X);
        const double t = Fp::mul_data(Fp::d(Y), tw(w, ws), Fp::d(P.fp_p), Fp::d(P.fp_pinv));
        X = Fp::u(__dadd_rn(x, t));
        Y = Fp::u(__dadd_rn(x, -t));
    }
    template <bool UNI, bool FIRST = false> static __device__ __forceinline__ void inv(uint64_t &X, uint64_t &Y, uint32_t w, uint32_t ws, const P32 &P) {
        const double x = Fp::d(X), y = Fp::d(Y);
        X = Fp::u(__dadd_rn(x, y));
        Y = Fp::u(Fp::mul_data(__dadd_rn(x, -y), tw(w, ws), Fp::d(P.fp_p), Fp::d(P.fp_pinv)));
    }
    static __device__ __forceinline__ void inv_norm(uint64_t &X, uint64_t &Y, const P32 &P) {
        const double x = Fp::d(X), y = Fp::d(Y), p = Fp::d(P.fp_p), pinv = Fp::d(P.fp_pinv);
        X = Fp::u(Fp::mul_data(__dadd_rn(x, y), Fp::d(P.fp_n_inv), p, pinv));
        Y = Fp::u(Fp::mul_data(__dadd_rn(x, -y), Fp::d(P.fp_last_w), p, pinv));
    }
};
// the word-level hooks of the class (the butterflies live in BoxOps<CLS_FPW>: the kernels never run them on bare words)
template <> struct Bfly<uint32_t, CLS_FPW> {
    using T = uint32_t;
    static constexpr bool IS_FP = false;      // no in-transform range reductions (see above)
    static constexpr bool FUSED_LAZY = true;  // the fused kernels' pointwise steps take the centred int32 pattern
    // canonical word -> centred residue as an int32 pattern
    static __device__ __forceinline__ T load_fix(T v, const ModParams<T> &P) { return v > (P.p >> 1) ? v - P.p : v; }
    static __device__ __forceinline__ T reduce(T v, const ModParams<T> &) { return v; }
    static __device__ __forceinline__ T pre_inverse(T v, const ModParams<T> &) { return v; }
    // centred int32 pattern -> canonical word
    static __device__ __forceinline__ T finish_fwd(T v, const ModParams<T> &P) {
        return v + (P.p & (T)((int32_t)v >> 31));
    }
    static __device__ __forceinline__ T finish_inv(T v, const ModParams<T> &P) { return finish_fwd(v, P); }
};

// ---------------------------------------------------------------------------------------------
// CLS_PM64: p = 2^64 - c, c < 2^32 (64-bit words)
//
// Registers hold ANY representative in [0, 2^64) of a residue; products by a canonical factor come out canonical.
//   product   y w = H 2^64 + L (the carry-out trick of shoup_core gives H; L = lo32(y0 w0) | lo32(...) << 32)
//             = L + H c                      (2^64 = c)
//             = L + H0 c + (H1 c) 2^32       three 32-bit limbs and a top word  top = hi32(H1 c) + carries <= c + 1
//             = (limbs) + top c              one more multiply-add, at most one further carry (+ c)
//             w < p bounds H <= p - 2, which keeps `top` inside 32 bits even for c = 2^32 - 1 (then H1 <= 2^32 - 2).
//   sum / difference with a canonical second operand: one wrap at most, corrected by +-c (2^64 = c).  The forward
//             transform leaves any representative (its x role only ever meets canonical products); the inverse keeps
//             everything canonical (sums of two canonical values: one conditional -p).
// The reference computes exact `%` products here (src/prime64/generic_solinas.rs:42-128): same residues.
// One inline-asm block (fixed scratch v[2:7], two wait states behind every VALU-written SGPR, see shoup_core).
// ---------------------------------------------------------------------------------------------
#define CNTT_PM_BODY                                          \
    "v_mad_u64_u32 v[2:3], %[k], %[y0], %[w1], %[u]\n\t"      \
    "v_mul_lo_u32 v6, %[y0], %[w0]\n\t"                       \
    "v_mov_b32 v7, v2\n\t"                                    \
    "v_mov_b32 v2, v3\n\t"                                    \
    "v_cndmask_b32_e64 v3, 0, 1, %[k]\n\t"                    \
    "v_mad_u64_u32 v[2:3], vcc, %[y1], %[w1], v[2:3]\n\t"     \
    "v_mad_u64_u32 v[6:7], %[k], v2, %[c], v[6:7]\n\t"        \
    "v_mad_u64_u32 v[4:5], vcc, v3, %[c], 0\n\t"              \
    "v_add_co_u32 v7, vcc, v7, v4\n\t"                        \
    "s_nop 1\n\t"                                             \
    "v_addc_co_u32 v5, vcc, 0, v5, vcc\n\t"                   \
    "v_addc_co_u32_e64 v5, vcc, v5, 0, %[k]\n\t"              \
    "v_mad_u64_u32 v[6:7], %[k], v5, %[c], v[6:7]\n\t"        \
    "s_nop 1\n\t"                                             \
    "v_cndmask_b32_e64 v4, 0, 1, %[k]\n\t"                    \
    "v_mad_u64_u32 v[6:7], vcc, v4, %[c], v[6:7]"

// y * w mod (2^64 - c) for any y < 2^64 and w < p: some representative in [0, 2^64)
template <bool UNI> __device__ __forceinline__ uint64_t pm_mul_lazy(uint64_t y, uint64_t w, uint32_t c) {
    const uint32_t y0 = (uint32_t)y, y1 = (uint32_t)(y >> 32);
    const uint32_t w0 = (uint32_t)w, w1 = (uint32_t)(w >> 32);
    const uint64_t t = __umulhi(y0, w0);
    const uint64_t u = (uint64_t)y1 * w0 + t;  // cannot overflow
    uint64_t out, ck;
    if constexpr (UNI)
        asm(CNTT_PM_BODY : "=&{v[6:7]}"(out), [k] "=&s"(ck)
            : [y0] "v"(y0), [y1] "v"(y1), [w0] "s"(w0), [w1] "s"(w1), [c] "s"(c), [u] "v"(u)
            : "vcc", "v2", "v3", "v4", "v5");
    else
        asm(CNTT_PM_BODY : "=&{v[6:7]}"(out), [k] "=&s"(ck)
            : [y0] "v"(y0), [y1] "v"(y1), [w0] "v"(w0), [w1] "v"(w1), [c] "s"(c), [u] "v"(u)
            : "vcc", "v2", "v3", "v4", "v5");
    return out;
}

template <class T> struct Bfly<T, CLS_PM64> {
    static_assert(sizeof(T) == 8, "CLS_PM64 is a 64-bit class");
    static constexpr bool IS_FP = false;
    static constexpr bool FUSED_LAZY = true;  // products take any representative: fused kernels skip finish_fwd
    static __device__ __forceinline__ T load_fix(T v, const ModParams<T> &) { return v; }
    static __device__ __forceinline__ T reduce(T v, const ModParams<T> &) { return v; }
    static __device__ __forceinline__ T canon(T v, const ModParams<T> &P) { return v >= P.p ? v + P.pm_c : v; }  // v - p
    template <bool UNI> static __device__ __forceinline__ T mulc(T y, T w, const ModParams<T> &P) {
        return canon(pm_mul_lazy<UNI>(y, w, P.pm_c), P);
    }
    // x + t, x - t for any x and canonical t
    static __device__ __forceinline__ T add_c(T x, T t, const ModParams<T> &P) {
        const T s = x + t;
        return s + (s < x ? (T)P.pm_c : (T)0);
    }
    static __device__ __forceinline__ T sub_c(T x, T t, const ModParams<T> &P) {
        const T d = x - t;
        return d - (x < t ? (T)P.pm_c : (T)0);
    }
    template <bool UNI = false, bool FIRST = false>
    static __device__ __forceinline__ void fwd(T &x, T &y, T w, T, const ModParams<T> &P) {
        const T t = mulc<UNI>(y, w, P);
        y = sub_c(x, t, P);
        x = add_c(x, t, P);
    }
    // The inverse keeps every value CANONICAL (its inputs are: memory words, products, or pre_inverse()d accumulators):
    // a canonical sum is one add, two compares and a conditional +c; the difference wraps at most once.
    static __device__ __forceinline__ T add_canon(T x, T y, const ModParams<T> &P) {
        const T s = x + y;
        const bool k = (s < x) | (s >= P.p);           // passed 2^64, or landed in [p, 2^64)
        return s + (k ? (T)P.pm_c : (T)0);             // - p
    }
    template <bool UNI = false, bool FIRST = false>
    static __device__ __forceinline__ void inv(T &x, T &y, T w, T, const ModParams<T> &P) {
        const T d = sub_c(x, y, P);
        x = add_canon(x, y, P);
        y = mulc<UNI>(d, w, P);
    }
    static __device__ __forceinline__ void inv_norm(T &x, T &y, const ModParams<T> &P) {
        const T d = sub_c(x, y, P), s = add_c(x, y, P);   // both go straight into products: any representative will do
        x = mulc<true>(s, P.pm_n_inv, P);
        y = mulc<true>(d, P.pm_last_w, P);
    }
    static __device__ __forceinline__ T pre_inverse(T v, const ModParams<T> &P) { return canon(v, P); }
    static __device__ __forceinline__ T finish_fwd(T v, const ModParams<T> &P) { return canon(v, P); }
    static __device__ __forceinline__ T finish_inv(T v, const ModParams<T> &) { return v; }  // canonical already
};

// ---------------------------------------------------------------------------------------------
// pointwise kernels' arithmetic (src/prime64.rs:534-584,690-699; src/prime32.rs:383-408,...)
// ---------------------------------------------------------------------------------------------
template <class T> struct Wide;
template <> struct Wide<uint32_t> {
    static __device__ __forceinline__ void mul(uint32_t a, uint32_t b, uint32_t &lo, uint32_t &hi) {
        const uint64_t d = (uint64_t)a * b;
        lo = (uint32_t)d;
        hi = (uint32_t)(d >> 32);
    }
};
template <> struct Wide<uint64_t> {
    static __device__ __forceinline__ void mul(uint64_t a, uint64_t b, uint64_t &lo, uint64_t &hi) {
        lo = a * b;
        hi = mulhi(a, b);
    }
};

// a*b mod p in [0, 2p)  (Barrett as the reference: c1 = d >> (Q-1), c3 = hi(c1 * p_barrett))
template <class T> __device__ __forceinline__ T barrett_mul_lazy(T a, T b, const ModParams<T> &P) {
    constexpr int B = sizeof(T) * 8;
    T lo, hi;
    Wide<T>::mul(a, b, lo, hi);
    const uint32_t sh = P.big_q - 1;  // 1 <= sh < B  (p >= 2 ... p < 2^(B-1))
    const T c1 = (T)((lo >> sh) | (hi << (B - sh)));
    const T c3 = mulhi(c1, P.p_barrett);
    return lo - P.p * c3;
}

template <class T> __device__ __forceinline__ T mul_normalize(T a, T b, const ModParams<T> &P, bool generic) {
    if (generic) {
        // mont(a, b) = a b R^-1 ; mont(., n_inv R^2) = a b n_inv   (n_inv field holds n_inv * R^2 mod p)
        const T t = mont_mul(a, b, P.p, P.pinv_neg);
        return mont_mul(t, P.n_inv, P.p, P.pinv_neg);
    }
    const T prod = barrett_mul_lazy<T>(a, b, P);
    const T t = shoup_mul<T, true>(prod, P.n_inv, P.n_inv_shoup, P.neg_p);
    return csub<T>(t, P.p);
}
// a*b in the range the inverse butterflies of class CLS accept, WITHOUT the 1/N factor (Bfly::inv_norm applies it):
// LAZY [0, 2p); STRICT canonical; GENERIC a b / R canonical (inv_norm's constants carry the R^2).
template <class T, int CLS> __device__ __forceinline__ T mul_for_inv(T a, T b, const ModParams<T> &P) {
    if constexpr (CLS == CLS_PM64) {
        return Bfly<T, CLS>::template mulc<false>(a, b, P);  // a: any representative, b: canonical word from memory
    } else if constexpr (is_fp_class(CLS)) {
        // a: the forward transform's lazy double (|a| < 2^53), b: a canonical word from memory; result |.| <= 0.875 p
        const double p = Fp::d(P.fp_p), pinv = Fp::d(P.fp_pinv);
        return Fp::u(Fp::mul_data(Fp::reduce(Fp::d(a), p, pinv), Fp::from_word(b), p, pinv));
    } else if constexpr (CLS == CLS_FPW) {
        // a: centred int32 pattern left by the forward transform, b: a canonical word from memory; result: centred pattern
        const double p = Fp::d(P.fp_p), pinv = Fp::d(P.fp_pinv);
        const double t = Fp::mul_data((double)(int32_t)a, (double)b, p, pinv);  // |t| <= (1/2 + 2^-20) p: reduce once more
        return (T)(int32_t)Fp::reduce(t, p, pinv);
    } else if constexpr (CLS == CLS_GENERIC) {
        return mont_mul(a, b, P.p, P.pinv_neg);
    } else if constexpr (CLS == CLS_STRICT) {
        // the low B bits of a remainder estimate in [0, 3p) (barrett_mul_lazy, as the reference: whatever that word is, its residue is what the
        // reference multiplies by 1/N): two conditional subtractions make it canonical, which this class's inverse butterflies assume
        return csub<T>(csub<T>(barrett_mul_lazy<T>(a, b, P), P.p), P.p);
    } else {
        return barrett_mul_lazy<T>(a, b, P);
    }
}
// ---------------------------------------------------------------------------------------------
// The pointwise product INSIDE the fused product kernels (forward transform -> product -> inverse transform in registers; round 4).
// Lazy class: a is the forward transform's lazy output in [0, 4p) AS IT IS -- no canonicalisation, no conditional subtraction --, b a
// canonical word of rhs_ntt; a Montgomery product  u = (a b + m p) / 2^B,  m = lo(a b) (-p^-1) mod 2^B:
//     a < 4p < 2^B, b <= p - 1  =>  hi(a b) <= p - 1,  hi(m p) <= p - 1,  u = hi(a b) + hi(m p) + [lo(a b) != 0] <= 2p - 1
// (lo(a b) + lo(m p) = 0 mod 2^B, so the carry into the upper word is 1 exactly when lo(a b) != 0).  u = a b 2^-B mod p goes straight
// into the inverse transform, whose last stage -- the one that applies 1/N (Bfly::inv_norm) -- takes constants multiplied by 2^B
// (mul_inv_params): 20 instructions instead of 8 (canonicalisation) + 26 (Barrett, mul_for_inv) for 64-bit words, 3 instead of 9 for
// 32-bit ones.  Every other class keeps mul_for_inv.
// ---------------------------------------------------------------------------------------------
// NOT the strict class (2^(B-2) <= p < 2^(B-1)), although the same product would be valid there (a < 2p <= 2^B): the reference's Barrett
// product (src/prime32.rs mul_assign_normalize / mul_accumulate, src/prime64.rs likewise) leaves d - c3 p in [0, 3p) and keeps its low
// B bits; for p > 2^B / 3 that remainder can pass 2^B, and the reference's result is then off by 2^B mod p (measured: 7e-4 of the products
// for p = 2127586817, 5e-6 for p = 1896656897).  barrett_mul_lazy reproduces that bit for bit, an exact product does not
// (tests/test_gpu_parity.py::test_strict_class_keeps_the_reference_barrett_wrap), so the strict class keeps mul_for_inv / mul_acc.
template <int CLS> __host__ __device__ constexpr bool mul_is_mont() { return CLS == CLS_LAZY; }
// FIN of the forward transform in front of mul_fused: does it have to canonicalise its outputs?
template <class T, int CLS> __host__ __device__ constexpr bool mul_fwd_fin() { return !Bfly<T, CLS>::FUSED_LAZY && !mul_is_mont<CLS>(); }
template <class T, int CLS> __device__ __forceinline__ T mul_fused(T a, T b, const ModParams<T> &P) {
    if constexpr (mul_is_mont<CLS>()) {
        T lo, hi;
        Wide<T>::mul(a, b, lo, hi);
        const T m = lo * P.pinv_neg;
        return hi + mulhi(m, P.p) + (lo != 0 ? (T)1 : (T)0);
    } else {
        return mul_for_inv<T, CLS>(a, b, P);
    }
}
// an accumulator of the fused chains -> what the inverse transform's first stage accepts (Montgomery classes: times 2^B, in [0, 2p))
template <class T, int CLS> __device__ __forceinline__ T chain_pre_inverse(T v, const ModParams<T> &P) {
    if constexpr (mul_is_mont<CLS>()) return shoup_mul<T, true>(v, P.mont_r, P.mont_r_shoup, P.neg_p);
    else return Bfly<T, CLS>::pre_inverse(v, P);
}
// the parameters the inverse half of a fused product kernel runs on
template <class T, int CLS> __device__ __forceinline__ ModParams<T> mul_inv_params(const ModParams<T> &P) {
    ModParams<T> Q = P;
    if constexpr (mul_is_mont<CLS>()) {
        Q.n_inv = P.mont_n_inv;
        Q.n_inv_shoup = P.mont_n_inv_shoup;
        Q.last_w = P.mont_last_w;
        Q.last_w_shoup = P.mont_last_w_shoup;
    }
    return Q;
}

template <class T> __device__ __forceinline__ T normalize1(T a, const ModParams<T> &P, bool generic) {
    if (generic) {
        // n_inv field = n_inv * R^2 ; mont(a, R^-1-free) : mont(a, n_inv R^2) = a n_inv R ; one more REDC by 1
        const T t = mont_mul(a, P.n_inv, P.p, P.pinv_neg);
        return mont_mul(t, (T)1, P.p, P.pinv_neg);
    }
    const T t = shoup_mul<T, true>(a, P.n_inv, P.n_inv_shoup, P.neg_p);
    return csub<T>(t, P.p);
}
template <class T> __device__ __forceinline__ T mul_acc(T acc, T a, T b, const ModParams<T> &P, bool generic) {
    if (generic) {
        const T t = mont_mul(a, b, P.p, P.pinv_neg);  // a b R^-1
        const T prod = mont_mul(t, P.r2, P.p, P.pinv_neg);
        return add_mod<T>(acc, prod, P.p);
    }
    T prod = barrett_mul_lazy<T>(a, b, P);
    prod = csub<T>(prod, P.p);
    return csub<T>(prod + acc, P.p);
}

// acc + a * b in the accumulator form of class CLS (the fused mul_accumulate chains): the integer classes keep
// canonical accumulators; the double classes add the product (|.| <= 0.875 p, `a` already range-reduced) to a lazy double.
template <class T, int CLS> __device__ __forceinline__ T mul_acc_cls(T acc, T a, T b, const ModParams<T> &P) {
    if constexpr (mul_is_mont<CLS>()) {
        // round 4: a is the forward transform's lazy output (no canonicalisation), the product a Montgomery product u < 2p (mul_fused),
        // the accumulator lives in [0, 2p): one conditional subtraction of 2p per term.  The factor 2^-B common to every term is undone
        // once per output coefficient in front of the inverse transform (chain_pre_inverse): 25 instructions per term, output and
        // coefficient instead of 8 / J + 35 for 64-bit words, 6 instead of 10 for 32-bit ones.
        T lo, hi;
        Wide<T>::mul(a, b, lo, hi);
        const T m = lo * P.pinv_neg;
        const T u = hi + mulhi(m, P.p) + (lo != 0 ? (T)1 : (T)0);
        return csub_two_p<T>(acc + u, P.two_p, P.neg_two_p);   // 4p < 2^B: no wrap
    } else if constexpr (CLS == CLS_PM64) {
        return Bfly<T, CLS>::add_c(acc, Bfly<T, CLS>::template mulc<false>(a, b, P), P);
    } else if constexpr (is_fp_class(CLS)) {
        const double p = Fp::d(P.fp_p), pinv = Fp::d(P.fp_pinv);
        return Fp::u(__dadd_rn(Fp::d(acc), Fp::mul_data(Fp::d(a), Fp::from_word(b), p, pinv)));
    } else if constexpr (CLS == CLS_FPW) {  // accumulator, a: centred int32 patterns; b: canonical word
        const double p = Fp::d(P.fp_p), pinv = Fp::d(P.fp_pinv);
        const double t = Fp::mul_data((double)(int32_t)a, (double)b, p, pinv);
        return (T)(int32_t)Fp::reduce(__dadd_rn((double)(int32_t)acc, t), p, pinv);
    } else {
        return mul_acc<T>(acc, a, b, P, CLS == CLS_GENERIC);
    }
}

}  // namespace cntt
