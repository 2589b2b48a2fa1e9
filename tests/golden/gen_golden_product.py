#!/usr/bin/env python3
"""Golden vectors for product::Plan (src/product.rs), generated with Python big integers only.

Independent of oracle/ and of the HIP code: per-prime transforms come from gen_golden.Plan (textbook
Cooley-Tukey / Gentleman-Sande on Python ints), the recombination is a textbook CRT sum (not Garner),
the residue split is `%`.  The cases are the reference's own tests (src/product.rs:976-1166: u64x1,
u32x1, u32x2, u30x2, u32x4, u32x2_u64x1) plus FwdMode::Bounded and the pointwise calls.

    python3 tests/golden/gen_golden_product.py      # rewrites tests/golden/golden_product_v1.json
"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from gen_golden import Plan, fill_u64, is_prime, negacyclic, digest  # noqa: E402


def largest_prime(factor, offset, lo, hi):  # src/prime.rs:183-212 semantics: largest p = factor*k+offset in [lo, hi]
    k = (hi - offset) // factor
    while factor * k + offset >= lo and k >= 0:
        if is_prime(factor * k + offset):
            return factor * k + offset
        k -= 1
    return None


def crt(residues, primes):
    big = 1
    for p in primes:
        big *= p
    x = 0
    for r, p in zip(residues, primes):
        m = big // p
        x += r * m * pow(m, -1, p)
    return x % big


def pack_domain(per_prime, primes, n):
    """src/product.rs:261-270: u32 residues two per u64 word (little endian), then the u64 residues."""
    words = []
    flat32 = [v for vals, p in zip(per_prime, primes) if p < 2**32 for v in vals]
    for i in range(0, len(flat32), 2):
        words.append(flat32[i] | (flat32[i + 1] << 32))
    for vals, p in zip(per_prime, primes):
        if p >= 2**32:
            words.extend(vals)
    return words


def main():
    out = {"version": 1, "generator": "tests/golden/gen_golden_product.py", "cases": [], "none": []}
    cases = []
    for n in (64, 256):
        f = 2 * n
        cases.append(("u64x1", n, [largest_prime(f, 1, 0, 2**64 - 1)]))
        cases.append(("u32x1", n, [largest_prime(f, 1, 0, 2**32 - 1)]))
        p0 = largest_prime(f, 1, 0, 2**32 - 1)
        cases.append(("u32x2", n, [p0, largest_prime(f, 1, 0, p0 - 1)]))
        p0 = largest_prime(f, 1, 0, 2**30)
        cases.append(("u30x2", n, [p0, largest_prime(f, 1, 0, p0 - 1)]))
        ps = [largest_prime(f, 1, 0, 2**16 - 1)]
        for _ in range(3):
            ps.append(largest_prime(f, 1, 0, ps[-1] - 1))
        cases.append(("u32x4", n, ps))
        p1 = largest_prime(f, 1, 0, 2**15)
        cases.append(("u32x2_u64x1", n, [largest_prime(f, 1, 0, 2**33), p1, largest_prime(f, 1, 0, p1 - 1)]))
    # beyond the reference's tests: two 64-bit-class primes, and a 1 among the factors (skipped, :170-171)
    cases.append(("u64x2_with_one", 64, [1, largest_prime(128, 1, 0, 2**32 + 2**20), largest_prime(128, 1, 0, 2**31)]))

    for name, n, given in cases:
        primes = sorted(p for p in given if p != 1)
        big = 1
        for p in primes:
            big *= p
        assert big < 2**64
        plans = [Plan(n, p, 32 if p < 2**32 else 64) for p in primes]
        seed = 1000 + len(out["cases"])
        a = fill_u64(n, big, seed)
        b = fill_u64(n, big, seed + 500)
        init = fill_u64(n, big, seed + 900)
        fa = [pl.fwd([x % pl.p for x in a]) for pl in plans]
        fb = [pl.fwd([x % pl.p for x in b]) for pl in plans]
        dom_a = pack_domain(fa, primes, n)
        ia = [pl.inv(v) for pl, v in zip(plans, fa)]
        replace = [crt([v[i] for v in ia], primes) for i in range(n)]
        assert replace == [x * n % big for x in a]
        accumulate = [(init[i] + replace[i]) % big for i in range(n)]
        # mul_assign_normalize: fa * fb * n^-1 per prime; inverse transform = negacyclic product mod big
        mn = [[x * y % pl.p * pl.n_inv % pl.p for x, y in zip(u, v)] for pl, u, v in zip(plans, fa, fb)]
        prod = [crt([pl.inv(v)[i] for pl, v in zip(plans, mn)], primes) for i in range(n)]
        assert prod == negacyclic(n, big, a, b)
        nrm = [[x * pl.n_inv % pl.p for x in u] for pl, u in zip(plans, fa)]
        macc = [[(x + y * z) % pl.p for x, y, z in zip(u, v, w)] for pl, u, v, w in zip(plans, fa, fa, fb)]
        ent = {"name": name, "n": n, "factors": [str(p) for p in given], "modulus": str(big),
               "n32": sum(p < 2**32 for p in primes), "n64": sum(p >= 2**32 for p in primes),
               "seed_a": seed, "seed_b": seed + 500, "seed_init": seed + 900,
               "fwd_sha256": digest(dom_a, 8), "inv_replace_sha256": digest(replace, 8),
               "inv_accumulate_sha256": digest(accumulate, 8),
               "mul_assign_normalize_sha256": digest(pack_domain(mn, primes, n), 8),
               "normalize_sha256": digest(pack_domain(nrm, primes, n), 8),
               "mul_accumulate_sha256": digest(pack_domain(macc, primes, n), 8),
               "polymul_sha256": digest(prod, 8)}
        if n == 64:
            ent["a"] = [str(x) for x in a]
            ent["fwd"] = [str(x) for x in dom_a]
            ent["inv_replace"] = [str(x) for x in replace]
        if len(primes) == 2 and primes[1] < 2**32:
            # FwdMode::Bounded(bound): inputs are centred values of magnitude < bound stored mod big
            bound = 1 << 20
            c = [int(x) - bound + 1 for x in fill_u64(n, 2 * bound - 1, seed + 77)]
            std = [x % big for x in c]
            fb_ = [pl.fwd([x % pl.p for x in c]) for pl in plans]
            ent["bounded"] = {"bound": bound, "seed": seed + 77, "standard_sha256": digest(std, 8),
                              "fwd_sha256": digest(pack_domain(fb_, primes, n), 8)}
        out["cases"].append(ent)

    p0, p1 = largest_prime(512, 1, 0, 2**33), largest_prime(512, 1, 0, 2**15)
    out["none"] = [
        {"why": "zero factor (test_plan_failure_zero)", "n": 256, "modulus": "0", "factors": [str(p0), "0"]},
        {"why": "duplicate factor (test_plan_failure_dup)", "n": 256, "modulus": str(p0 * p1 * p1 % 2**64),
         "factors": [str(p1), str(p0), str(p1)]},
        {"why": "odd size", "n": 255, "modulus": str(p0), "factors": [str(p0)]},
        {"why": "product != modulus", "n": 256, "modulus": str(p0 * p1 + 1), "factors": [str(p0), str(p1)]},
        {"why": "product overflows u64", "n": 256, "modulus": "1", "factors": [str(2**64 - 2**32 + 1), str(p0)]},
        {"why": "prime32 plan needs n >= 32", "n": 16, "modulus": str(p1), "factors": [str(p1)]},
        {"why": "composite factor", "n": 256, "modulus": str(p0 * 513), "factors": [str(p0), "513"]},
    ]
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden_product_v1.json")
    with open(path, "w") as f:
        json.dump(out, f, indent=0, separators=(",", ":"))
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
