/* Counterpart of the reference's examples/mul_poly_native.rs on the C ABI: negacyclic product of two u32
 * polynomials modulo 2^32 (wrapping), schoolbook versus native32::Plan32::negacyclic_polymul. */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "../include/cntt.h"

static uint64_t rng_state = 0x243F6A8885A308D3ull;
static uint32_t next_u32(void) {
    uint64_t z = (rng_state += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return (uint32_t)((z ^ (z >> 31)) >> 32);
}

int main(void) {
    const size_t n = 1024;
    uint32_t *lhs = malloc(n * 4), *rhs = malloc(n * 4), *want = calloc(n, 4), *prod = calloc(n, 4);
    uint32_t *full = calloc(2 * n, 4);
    for (size_t i = 0; i < n; ++i) lhs[i] = next_u32();
    for (size_t i = 0; i < n; ++i) rhs[i] = next_u32();
    for (size_t i = 0; i < n; ++i)
        for (size_t j = 0; j < n; ++j) full[i + j] += lhs[i] * rhs[j]; /* wrapping */
    for (size_t i = 0; i < n; ++i) want[i] = full[i] - full[n + i];

    cntt_native_t *plan = NULL;
    int rc = cntt_native_plan_new(CNTT_NATIVE32_PLAN32, n, &plan);
    if (rc == CNTT_OK) rc = cntt_native_negacyclic_polymul(plan, prod, n, lhs, n, rhs, n);
    if (rc != CNTT_OK) {
        fprintf(stderr, "status %d: %s\n", rc, cntt_last_error());
        return 1;
    }
    cntt_native_plan_free(plan);
    if (memcmp(prod, want, n * 4) != 0) {
        fprintf(stderr, "MISMATCH\n");
        return 2;
    }
    printf("Success!\n");
    free(lhs), free(rhs), free(want), free(prod), free(full);
    return 0;
}
