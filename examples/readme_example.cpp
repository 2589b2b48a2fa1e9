// The crate's README / doc-test example (README.md:41-57, src/lib.rs:25-49) and a product::Plan round trip
// (src/product.rs:1029-1059) through include/cntt.hpp, the C++17 mirror of the API over the C ABI.
#include <cstdio>
#include <vector>

#include "../include/cntt.hpp"

__extension__ typedef unsigned __int128 u128;

int main() {
    try {
        const uint32_t p = 1062862849;
        const size_t n = 32;
        auto plan = cntt::prime32::Plan::try_new(n, p).value();
        std::vector<uint32_t> data(n), buf(n);
        for (size_t i = 0; i < n; ++i) data[i] = buf[i] = (uint32_t)i;
        plan.fwd(buf);
        plan.inv(buf);
        for (size_t i = 0; i < n; ++i)
            if (buf[i] != (uint32_t)((uint64_t)data[i] * n % p)) return std::puts("MISMATCH prime32"), 2;

        // Plan::try_new returns None for a composite modulus; a wrong slice length panics
        if (cntt::prime32::Plan::try_new(n, p + 2)) return std::puts("expected None"), 2;
        bool panicked = false;
        try {
            std::vector<uint32_t> shorter(n - 1);
            plan.fwd(shorter);
        } catch (const cntt::Panic &) {
            panicked = true;
        }
        if (!panicked) return std::puts("expected a panic"), 2;

        // product::Plan with two 32-bit primes: inv(fwd(x)) = n * x mod p0 * p1
        const uint64_t p0 = 4294957057ull, p1 = 4294962689ull;  // src/product.rs:1033-1035 at n = 256
        const size_t m = 256;
        auto pplan = cntt::product::Plan::try_new(m, p0 * p1, {p0, p1}).value();
        std::vector<uint64_t> standard(m), ntt(pplan.ntt_domain_len()), round(m);
        for (size_t i = 0; i < m; ++i) standard[i] = (0x9E3779B97F4A7C15ull * (i + 1)) % (p0 * p1);
        pplan.fwd(ntt, standard, cntt::product::FwdMode::Generic());
        pplan.inv(round, ntt, cntt::product::InvMode::Replace);
        for (size_t i = 0; i < m; ++i)
            if (round[i] != (uint64_t)((u128)standard[i] * m % (p0 * p1))) return std::puts("MISMATCH product"), 2;
        // host-only helpers of the mirror: the batch partition for multi-device callers and the testing-only switchboard
        const auto sh = cntt::shard_bounds(10, 3, 2);   // 10 polynomials over 3 devices: [0,4) [4,7) [7,10)
        if (sh.first != 7 || sh.second != 10 || cntt::debug_get("native_acc") != 1) return std::puts("MISMATCH helpers"), 2;
        std::puts("Success!");
        return 0;
    } catch (const cntt::DeviceError &e) {
        std::fprintf(stderr, "device error (no CPU path): %s\n", e.what());
        return 1;
    }
}
