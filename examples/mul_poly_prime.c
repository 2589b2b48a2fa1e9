/* Counterpart of the reference's examples/mul_poly_prime.rs on the C ABI of libcntt_hip.so (include/cntt.h):
 * negacyclic product of two polynomials modulo a 32-bit NTT prime, schoolbook versus
 * fwd / fwd / mul_assign_normalize / inv.  Plain C: the boundary has no C++ or torch types.
 *   make -C examples && ./examples/mul_poly_prime        (needs a GPU: the library has no CPU path) */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "../include/cntt.h"

static uint64_t rng_state = 0x9E3779B97F4A7C15ull;
static uint32_t next_u32(void) { /* splitmix64 */
    uint64_t z = (rng_state += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return (uint32_t)((z ^ (z >> 31)) >> 32);
}

#define CHECK(call)                                                                        \
    do {                                                                                   \
        int rc_ = (call);                                                                  \
        if (rc_ != CNTT_OK) {                                                              \
            fprintf(stderr, "%s failed: status %d: %s\n", #call, rc_, cntt_last_error());  \
            return 1;                                                                      \
        }                                                                                  \
    } while (0)

int main(void) {
    const uint32_t p = 1073479681u; /* examples/mul_poly_prime.rs:6 */
    const size_t n = 1024;
    uint32_t *lhs = malloc(n * 4), *rhs = malloc(n * 4), *want = calloc(n, 4);
    uint32_t *full = calloc(2 * n, 4);
    for (size_t i = 0; i < n; ++i) lhs[i] = next_u32() % p;
    for (size_t i = 0; i < n; ++i) rhs[i] = next_u32() % p;

    /* method 1: schoolbook */
    for (size_t i = 0; i < n; ++i)
        for (size_t j = 0; j < n; ++j)
            full[i + j] = (uint32_t)(((uint64_t)full[i + j] + (uint64_t)lhs[i] * rhs[j] % p) % p);
    for (size_t i = 0; i < n; ++i) want[i] = (uint32_t)(((uint64_t)full[i] + (p - full[n + i])) % p);

    /* method 2: NTT on the GPU */
    cntt_plan32_t *plan = NULL;
    CHECK(cntt_prime32_plan_new(n, p, &plan));
    CHECK(cntt_prime32_fwd(plan, lhs, n));
    CHECK(cntt_prime32_fwd(plan, rhs, n));
    CHECK(cntt_prime32_mul_assign_normalize(plan, lhs, n, rhs, n));
    CHECK(cntt_prime32_inv(plan, lhs, n));
    cntt_prime32_plan_free(plan);

    if (memcmp(lhs, want, n * 4) != 0) {
        fprintf(stderr, "MISMATCH\n");
        return 2;
    }
    printf("Success!\n");
    free(lhs), free(rhs), free(want), free(full);
    return 0;
}
