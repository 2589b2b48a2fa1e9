#!/usr/bin/env python3
"""Generate tests/golden/golden_v1.json.

An INDEPENDENT pure-Python (big-int) restatement of the parts of concrete-ntt that determine
every output bit, used to pin the C oracle (oracle/cntt_oracle.c) and, through it, the HIP path.
It shares no code with the oracle: plain `%` arithmetic, an iterative (not recursive) schedule,
and canonical values everywhere (the reference's lazy [0,2p)/[0,4p) ranges are an internal detail;
all public outputs are canonical, SURVEY.md point 3).

What is followed literally (paths relative to /root/reference/):
  * root choice:   src/roots.rs:6-91 (Tonelli-Shanks chain from p-1, z = smallest non-residue)
  * table layout:  src/prime64.rs:183-218, src/prime32.rs:248-282, bit_rev src/lib.rs:118-121
  * plan scalars:  src/prime64.rs:752-756, src/prime32.rs:664-668
  * fwd schedule:  src/prime64/shoup.rs:544-706 == stage s, block i uses twid[2^s + i]
  * inv schedule:  src/prime64/shoup.rs:1306-1468 == m blocks use inv_twid[m + i], no 1/N
  * CRT:           src/native32.rs:28-56, src/native64.rs:91-141, src/native128.rs:20-118,
                   src/native_binary{32,64,128}.rs, constants src/lib.rs:447-652
Inputs come from splitmix64 (the same generator the GPU fill kernel and the oracle implement),
so fixtures only store seeds, not inputs.

Run:  python3 tests/golden/gen_golden.py   (takes ~1 minute; no dependency on the reference tree)
"""
import hashlib
import json
import os
import struct

M64 = (1 << 64) - 1


def splitmix64(x):
    x = (x + 0x9E3779B97F4A7C15) & M64
    x = ((x ^ (x >> 30)) * 0xBF58476D1CE4E5B9) & M64
    x = ((x ^ (x >> 27)) * 0x94D049BB133111EB) & M64
    return x ^ (x >> 31)


def fill_u64(count, bound, seed):
    out = []
    for i in range(count):
        r = splitmix64((seed + i) & M64)
        out.append((r * bound) >> 64 if bound else r)
    return out


def fill_u32(count, bound, seed):
    out = []
    for i in range(count):
        r = splitmix64((seed + i) & M64) >> 32
        out.append((r * bound) >> 32 if bound else r)
    return out


def bit_rev(nbits, i):
    return int(format(i, "0{}b".format(nbits))[::-1], 2) if nbits else 0


def is_prime(n):
    if n < 2:
        return False
    for q in (2, 3, 5, 7, 11, 13, 17, 19, 23, 29, 31, 37):
        if n % q == 0:
            return n == q
    d, s = n - 1, 0
    while d % 2 == 0:
        d //= 2
        s += 1
    for a in (2, 3, 5, 7, 11, 13, 17, 19, 23, 29, 31, 37):
        x = pow(a, d, n)
        if x in (1, n - 1):
            continue
        for _ in range(s - 1):
            x = x * x % n
            if x == n - 1:
                break
        else:
            return False
    return True


def get_z(p):
    n = 2
    while n < p:
        if pow(n, (p - 1) // 2, p) == p - 1:
            return n
        n += 1
    return None


def sqrt_mod_ex(p, q, s, z, n):
    m, c, t, r = s, pow(z, q, p), pow(n, q, p), pow(n, (q + 1) // 2, p)
    while True:
        if t == 0:
            return 0
        if t == 1:
            return r
        i, t_pow = 0, t
        while i < m:
            t_pow = t_pow * t_pow % p
            i += 1
            if t_pow == 1:
                break
        if i == m:
            return None
        b = pow(c, 1 << (m - i - 1), p)
        m, c = i, b * b % p
        t, r = t * c % p, r * b % p


def find_primitive_root(p, degree):
    n = degree.bit_length() - 1
    root = p - 1
    q, s = p - 1, 0
    while q % 2 == 0:
        q //= 2
        s += 1
    z = get_z(p)
    if z is None:
        return None
    for _ in range(n - 1):
        root = sqrt_mod_ex(p, q, s, z, root)
        if root is None:
            return None
    return root


class Plan:
    def __init__(self, n, p, bits):
        self.n, self.p, self.bits = n, p, bits
        w = find_primitive_root(p, 2 * n)
        self.w = w
        nbits = n.bit_length() - 1
        self.twid, self.inv_twid = [0] * n, [0] * n
        wk = 1
        for k in range(n):
            self.twid[bit_rev(nbits, k)] = wk
            self.inv_twid[bit_rev(nbits, (n - k) % n)] = wk if k == 0 else p - wk
            wk = wk * w % p
        self.has_shoup = p < (1 << (bits - 1))
        self.n_inv = pow(n, p - 2, p)
        self.n_inv_shoup = ((self.n_inv << bits) // p) & ((1 << bits) - 1)
        self.big_q = p.bit_length()
        self.p_barrett = ((1 << (self.big_q + bits - 1)) // p) & ((1 << bits) - 1)

    def shoup(self, x):
        return (x << self.bits) // self.p

    def fwd(self, a):
        a, n, p = list(a), self.n, self.p
        t, m = n, 1
        while m < n:
            t //= 2
            for i in range(m):
                w = self.twid[m + i]
                for j in range(2 * i * t, 2 * i * t + t):
                    u, v = a[j], a[j + t] * w % p
                    a[j], a[j + t] = (u + v) % p, (u - v) % p
            m *= 2
        return a

    def inv(self, a):
        a, n, p = list(a), self.n, self.p
        t, m = 1, n
        while m > 1:
            m //= 2
            for i in range(m):
                w = self.inv_twid[m + i]
                for j in range(2 * i * t, 2 * i * t + t):
                    u, v = a[j], a[j + t]
                    a[j], a[j + t] = (u + v) % p, (u - v) * w % p
            t *= 2
        return a


def try_new(n, p, bits):
    min_n = 16 if bits == 64 else 32
    if n < min_n or n & (n - 1) or not is_prime(p) or find_primitive_root(p, 2 * n) is None:
        return None
    return Plan(n, p, bits)


def negacyclic(n, mod, lhs, rhs):
    full = [0] * (2 * n)
    for i in range(n):
        if lhs[i] == 0:
            continue
        for j in range(n):
            full[i + j] += lhs[i] * rhs[j]
    return [(full[i] - full[i + n]) % mod for i in range(n)]


def digest(vals, word):
    fmt = {4: "<I", 8: "<Q"}.get(word)
    h = hashlib.sha256()
    for v in vals:
        h.update(struct.pack(fmt, v) if fmt else int(v).to_bytes(16, "little"))
    return h.hexdigest()


P32 = [0x3F5A0001, 0x3F5D0001, 0x3F760001, 0x3F820001, 0x3FAC0001,
       0x3FAF0001, 0x3FB10001, 0x3FBB0001, 0x3FDE0001, 0x3FFC0001]
P52 = [0x3FFFFFE770001, 0x3FFFFFEB90001, 0x3FFFFFEC80001,
       0x3FFFFFF8B0001, 0x3FFFFFFB80001, 0x3FFFFFFC70001]


def crt_mixed_radix(residues, primes, groups, wordbits):
    """Mixed-radix digits in the reference's grouping, centred lift decided by the TOP digit
    (e.g. src/native64.rs:127 `sign = v34 > P34/2`), wrapping to `wordbits`."""
    # residues modulo each group's product (exact CRT inside a group)
    gm, gr = [], []
    for g in groups:
        mod = 1
        for i in g:
            mod *= primes[i]
        x = 0
        for i in g:  # x = residue mod `mod`
            mi = mod // primes[i]
            x = (x + residues[i] * mi * pow(mi, -1, primes[i])) % mod
        gm.append(mod)
        gr.append(x)
    digits, prefix = [], 1
    value = 0
    for mod, r in zip(gm, gr):
        d = (r - value) * pow(prefix, -1, mod) % mod
        digits.append(d)
        value += d * prefix
        prefix *= mod
    sign = digits[-1] > gm[-1] // 2
    out = value - prefix if sign else value
    return out % (1 << wordbits)


NATIVE = {  # kind: (primes, groups, wordbits, binary)
    "native32_plan32": (P32[:3], [[0], [1], [2]], 32, False),
    "native64_plan32": (P32[:5], [[0], [1, 2], [3, 4]], 64, False),
    "native128_plan32": (P32[:10], [[0, 1], [2, 3], [4, 5], [6, 7], [8, 9]], 128, False),
    "native_binary32_plan32": (P32[:2], [[0], [1]], 32, True),
    "native_binary64_plan32": (P32[:3], [[0], [1], [2]], 64, True),
    "native_binary128_plan32": (P32[:5], [[0], [1, 2], [3, 4]], 128, True),
    "native32_plan52": (P52[:2], [[0], [1]], 32, False),
    "native64_plan52": (P52[:3], [[0], [1], [2]], 64, False),
    "native_binary32_plan52": (P52[:1], [[0]], 32, True),
    "native_binary64_plan52": (P52[:2], [[0], [1]], 64, True),
}


def main():
    out = {"version": 1, "generator": "tests/golden/gen_golden.py", "plans": [], "transforms": [],
           "pointwise": [], "crt": [], "polymul": [], "try_new_none": [], "prime_search": []}

    u64_primes = [1125899904679937, 2251799813554177, 4611686018427322369, 9223372036853661697,
                  18446744069414584321, 18446744073707716609]  # benches/ntt.rs:111-118
    u32_primes = [1062862849, 1073479681, 2147352577, 4293918721]  # README + benches/ntt.rs:87-91

    plan_cases = [(64, 16, p) for p in u64_primes] + [(64, 1024, p) for p in u64_primes] + \
                 [(32, 32, p) for p in u32_primes] + [(32, 1024, p) for p in u32_primes]
    for bits, n, p in plan_cases:
        pl = try_new(n, p, bits)
        ent = {"bits": bits, "n": n, "p": p, "z": get_z(p), "w": pl.w,
               "twid_head": pl.twid[:8], "inv_twid_head": pl.inv_twid[:8],
               "twid_sha256": digest(pl.twid, bits // 8), "inv_twid_sha256": digest(pl.inv_twid, bits // 8),
               "n_inv": pl.n_inv, "n_inv_shoup": pl.n_inv_shoup, "big_q": pl.big_q,
               "has_shoup": pl.has_shoup}
        if pl.has_shoup:
            ent["p_barrett"] = pl.p_barrett
            ent["twid_shoup_head"] = [pl.shoup(x) for x in pl.twid[:8]]
            ent["inv_twid_shoup_head"] = [pl.shoup(x) for x in pl.inv_twid[:8]]
            ent["twid_shoup_sha256"] = digest([pl.shoup(x) for x in pl.twid], bits // 8)
        out["plans"].append(ent)

    # transforms: iota input and seeded random input; full output for n <= 64, digest otherwise
    tcases = []
    for p in u64_primes:
        for n in (16, 32, 64, 256, 1024, 2048, 4096):
            tcases.append((64, n, p))
    tcases.append((64, 16384, 4611686018427322369))
    for p in u32_primes:
        for n in (32, 64, 512, 2048, 4096, 8192):
            tcases.append((32, n, p))
    for idx, (bits, n, p) in enumerate(tcases):
        pl = try_new(n, p, bits)
        seed = 0x5EED0000 + idx
        rnd = fill_u64(n, p, seed) if bits == 64 else fill_u32(n, p, seed)
        for name, data in (("iota", [i % p for i in range(n)]), ("splitmix", rnd)):
            f = pl.fwd(data)
            i_ = pl.inv(data)  # inverse transform applied to the same input (bit-reversed-order semantics)
            rt = pl.inv(f)
            assert rt == [x * n % p for x in data]
            ent = {"bits": bits, "n": n, "p": p, "input": name, "seed": seed,
                   "fwd_sha256": digest(f, bits // 8), "inv_sha256": digest(i_, bits // 8),
                   "fwd_head": f[:8], "inv_head": i_[:8]}
            if n <= 64:
                ent["fwd"], ent["inv"] = f, i_
            out["transforms"].append(ent)

    # pointwise: mul_assign_normalize, normalize, mul_accumulate on canonical inputs
    for idx, (bits, n, p) in enumerate([(64, 64, p) for p in u64_primes] + [(32, 64, p) for p in u32_primes]):
        pl = try_new(n, p, bits)
        seed = 0xABCD0000 + 16 * idx
        fill = fill_u64 if bits == 64 else fill_u32
        a, b, c = fill(n, p, seed), fill(n, p, seed + 1), fill(n, p, seed + 2)
        out["pointwise"].append({
            "bits": bits, "n": n, "p": p, "seed": seed,
            "mul_assign_normalize": [x * y % p * pl.n_inv % p for x, y in zip(a, b)],
            "normalize": [x * pl.n_inv % p for x in a],
            "mul_accumulate": [(z + x * y) % p for x, y, z in zip(a, b, c)],  # acc=c, lhs=a, rhs=b
        })

    # CRT on random residues (truth: mixed-radix digits with top-digit sign rule)
    for kind, (primes, groups, wbits, _b) in NATIVE.items():
        vecs = []
        for t in range(24):
            seed = 0xC0DE0000 + 97 * t
            res = [splitmix64(seed + i) % primes[i] for i in range(len(primes))]
            if t == 0:
                res = [0] * len(primes)
            if t == 1:
                res = [q - 1 for q in primes]
            vecs.append({"residues": res, "value": crt_mixed_radix(res, primes, groups, wbits)})
        out["crt"].append({"kind": kind, "vectors": vecs})

    # polymul: truth is the schoolbook wrapping convolution (independent of any root choice)
    for kind, (primes, groups, wbits, binary) in NATIVE.items():
        for n in (32, 64, 256):
            seed = 0xFACE0000 + n
            mask = (1 << wbits) - 1
            if wbits == 128:
                lhs = [(splitmix64(seed + 2 * i) << 64 | splitmix64(seed + 2 * i + 1)) & mask for i in range(n)]
                rhs = [(splitmix64(seed + 7777 + 2 * i) << 64 | splitmix64(seed + 7778 + 2 * i)) & mask
                       for i in range(n)]
            else:
                lhs = [splitmix64(seed + i) & mask for i in range(n)]
                rhs = [splitmix64(seed + 7777 + i) & mask for i in range(n)]
            if binary:
                rhs = [x & 1 for x in rhs]
            prod = negacyclic(n, 1 << wbits, lhs, rhs)
            ent = {"kind": kind, "n": n, "seed": seed, "wordbits": wbits, "binary": binary,
                   "prod_sha256": digest(prod, wbits // 8), "prod_head": [str(x) for x in prod[:4]]}
            if n == 32:
                ent["lhs"] = [str(x) for x in lhs]
                ent["rhs"] = [str(x) for x in rhs]
                ent["prod"] = [str(x) for x in prod]
            out["polymul"].append(ent)

    # construction failures (src/prime64.rs:709-713, :1879-1882, src/prime32.rs:635-640)
    out["try_new_none"] = [
        {"bits": 64, "n": 2048, "p": 1024},           # test_plan_crash_github_11
        {"bits": 64, "n": 8, "p": 4611686018427322369},     # n < 16
        {"bits": 64, "n": 48, "p": 4611686018427322369},    # not a power of two
        {"bits": 64, "n": 1 << 17, "p": 4611686018427322369},  # 2N does not divide p-1
        {"bits": 64, "n": 1024, "p": 4611686018427322371},  # composite
        {"bits": 32, "n": 16, "p": 1062862849},       # n < 32
        {"bits": 32, "n": 1 << 17, "p": 1062862849},  # P0 = 1 mod 2^17 only
        {"bits": 32, "n": 64, "p": 1062862851},       # composite
    ]
    for c in out["try_new_none"]:
        assert try_new(c["n"], c["p"], c["bits"]) is None, c
    # src/prime.rs:220-221 -- the only literal known answers in the reference
    out["prime_search"] = [
        {"factor": 6, "offset": 5, "lo": 0, "hi": M64, "value": 18446744073709551557},
        {"factor": 6, "offset": 1, "lo": 0, "hi": M64, "value": 18446744073709551427},
    ]

    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden_v1.json")
    with open(path, "w") as f:
        json.dump(out, f, indent=0, separators=(",", ":"))
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
