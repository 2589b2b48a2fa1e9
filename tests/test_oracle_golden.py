"""The C oracle against the committed golden vectors (independent big-int restatement,
tests/golden/gen_golden.py) and against the known answers of SURVEY.md 8(c)."""
import ctypes
import hashlib

import numpy as np
import pytest


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def _inputs(oracle, ent):
    n, p, bits = ent["n"], ent["p"], ent["bits"]
    dt = np.uint64 if bits == 64 else np.uint32
    if ent["input"] == "iota":
        return (np.arange(n, dtype=np.uint64) % np.uint64(p)).astype(dt)
    return oracle.fill_uniform(n, p, ent["seed"], bits)


def test_survey_known_answers(oracle):
    # README.md:41-57 / SURVEY.md 8(c): C1
    pl = oracle.Plan.try_new(32, 1062862849, 32)
    assert pl.table("twid")[:8].tolist() == [1, 1009014033, 706332808, 419281921, 642948907, 357582115,
                                             109014919, 58397221]
    assert pl.table("inv_twid")[:8].tolist() == [1, 53848816, 643580928, 356530041, 1004465628, 953847930,
                                                 705280734, 419913942]
    assert int(pl.table("twid_shoup")[1]) == 4077367345
    assert (pl.n_inv_mod_p, pl.n_inv_mod_p_shoup, pl.big_q, pl.p_barrett) == (1029648385, 4160749568, 30,
                                                                              2169464302)
    a = np.arange(32, dtype=np.uint32)
    pl.fwd(a)
    assert a.tolist() == [8337849, 878691898, 914453352, 923715776, 1012328021, 392768238, 897146226, 61013893,
                          417621120, 735327736, 679063422, 783376877, 515982175, 78075156, 1027473816, 272310227,
                          438592558, 299044371, 960039654, 1016112582, 163403347, 393949392, 147905694, 116938486,
                          994450562, 495302868, 148654772, 281181860, 737624634, 24323030, 192920823, 997675169]
    pl.inv(a)
    assert a.tolist() == [32 * i for i in range(32)]
    # C2 headline prime
    pl = oracle.Plan.try_new(1024, 4611686018427322369, 64)
    assert int(pl.table("twid")[1]) == 18014948273684224
    assert int(pl.table("twid_shoup")[1]) == 72059793094737920
    assert (pl.n_inv_mod_p, pl.n_inv_mod_p_shoup, pl.big_q, pl.p_barrett) == (
        4607182418799951937, 18428729675200069632, 62, 9223372036854906878)
    a = np.arange(1024, dtype=np.uint64)
    pl.fwd(a)
    assert a[:4].tolist() == [2200404270350297092, 1058654583805296332, 104808385409697427, 633088089304352372]


def test_plans(oracle, golden):
    for ent in golden["plans"]:
        pl = oracle.Plan.try_new(ent["n"], ent["p"], ent["bits"])
        assert pl is not None, ent
        L = oracle.lib()
        z, w = ctypes.c_uint64(0), ctypes.c_uint64(0)
        assert L.orc_get_z64(ent["p"], ctypes.addressof(z)) and z.value == ent["z"]
        assert L.orc_find_primitive_root64(ent["p"], 2 * ent["n"], ctypes.addressof(w)) and w.value == ent["w"]
        assert pl.table("twid")[:8].tolist() == ent["twid_head"]
        assert pl.table("inv_twid")[:8].tolist() == ent["inv_twid_head"]
        assert sha(pl.table("twid")) == ent["twid_sha256"]
        assert sha(pl.table("inv_twid")) == ent["inv_twid_sha256"]
        assert pl.n_inv_mod_p == ent["n_inv"] and pl.big_q == ent["big_q"]
        if ent["has_shoup"]:
            assert pl.n_inv_mod_p_shoup == ent["n_inv_shoup"] and pl.p_barrett == ent["p_barrett"]
            assert pl.table("twid_shoup")[:8].tolist() == ent["twid_shoup_head"]
            assert pl.table("inv_twid_shoup")[:8].tolist() == ent["inv_twid_shoup_head"]
            assert sha(pl.table("twid_shoup")) == ent["twid_shoup_sha256"]
        else:
            assert pl.table("twid_shoup") is None


def test_transforms(oracle, golden):
    plans = {}
    for ent in golden["transforms"]:
        key = (ent["bits"], ent["n"], ent["p"])
        if key not in plans:
            plans[key] = oracle.Plan.try_new(ent["n"], ent["p"], ent["bits"])
        pl = plans[key]
        x = _inputs(oracle, ent)
        f = x.copy()
        pl.fwd(f)
        assert f[:8].tolist() == ent["fwd_head"], key
        assert sha(f) == ent["fwd_sha256"], key
        assert int(f.max()) < ent["p"]
        i = x.copy()
        pl.inv(i)
        assert i[:8].tolist() == ent["inv_head"], key
        assert sha(i) == ent["inv_sha256"], key
        if "fwd" in ent:
            assert f.tolist() == ent["fwd"] and i.tolist() == ent["inv"]


def test_pointwise(oracle, golden):
    for ent in golden["pointwise"]:
        n, p, bits, seed = ent["n"], ent["p"], ent["bits"], ent["seed"]
        pl = oracle.Plan.try_new(n, p, bits)
        a, b, c = (oracle.fill_uniform(n, p, seed + k, bits) for k in range(3))
        x = a.copy()
        pl.mul_assign_normalize(x, b)
        assert x.tolist() == ent["mul_assign_normalize"]
        x = a.copy()
        pl.normalize(x)
        assert x.tolist() == ent["normalize"]
        acc = c.copy()
        pl.mul_accumulate(acc, a, b)
        assert acc.tolist() == ent["mul_accumulate"]


def _crt(oracle, kind, res):
    L = oracle.lib()
    if kind == "native32_plan32":
        return L.orc_reconstruct_32bit_012_u32(*res)
    if kind == "native_binary32_plan32":
        return L.orc_reconstruct_32bit_01(*res)
    if kind == "native_binary64_plan32":
        return L.orc_reconstruct_32bit_012_u64(*res)
    if kind == "native64_plan32":
        m = np.array(res, dtype=np.uint32)
        return L.orc_reconstruct_32bit_01234_v2_u64(m.ctypes.data)
    if kind == "native32_plan52":
        return L.orc_reconstruct_52bit_01_u32(*res)
    if kind == "native64_plan52":
        return L.orc_reconstruct_52bit_012(*res)
    if kind == "native_binary32_plan52":
        return L.orc_reconstruct_52bit_0(*res)
    if kind == "native_binary64_plan52":
        return L.orc_reconstruct_52bit_01_u64(*res)
    return None


def test_crt(oracle, golden):
    for grp in golden["crt"]:
        kind = grp["kind"]
        nat = oracle.Native.try_new(kind, 32)
        for v in grp["vectors"]:
            direct = _crt(oracle, kind, v["residues"])
            if direct is not None:
                assert direct == v["value"], (kind, v)
        # the 128-bit variants return u128 by value; exercise them through inv() on impulse data:
        # NTT-domain buffers whose inverse transform is N * residue at every index are out of reach
        # without the plan, so feed residues through fwd of a constant polynomial instead.
        if kind in ("native128_plan32", "native_binary128_plan32"):
            for v in grp["vectors"][:8]:
                res = nat.residues()
                for k, r in enumerate(v["residues"]):
                    pk = oracle.Plan.try_new(32, oracle.lib().orc_primes32_p(k), 32)
                    buf = np.zeros(32, dtype=np.uint32)
                    # constant term c: fwd gives c everywhere; choose c = r * n^-1 so that inv -> r at index 0
                    buf[0] = (r * pk.n_inv_mod_p) % pk.p
                    pk.fwd(buf)
                    res[k][:] = buf
                out = nat.words()
                nat.inv(out, res)
                got = int(out[0]) | (int(out[1]) << 64)
                assert got == v["value"], (kind, v)


def test_polymul(oracle, golden):
    for ent in golden["polymul"]:
        kind, n, seed, wb = ent["kind"], ent["n"], ent["seed"], ent["wordbits"]
        nat = oracle.Native.try_new(kind, n)
        assert nat is not None
        L = oracle.lib()
        if wb == 128:
            lhs = np.empty(2 * n, dtype=np.uint64)
            rhs = np.empty(2 * n, dtype=np.uint64)
            for i in range(n):  # u128 = hi << 64 | lo, stored little-endian (lo first)
                lhs[2 * i + 1], lhs[2 * i] = L.orc_splitmix64(seed + 2 * i), L.orc_splitmix64(seed + 2 * i + 1)
                rhs[2 * i + 1], rhs[2 * i] = (L.orc_splitmix64(seed + 7777 + 2 * i),
                                              L.orc_splitmix64(seed + 7778 + 2 * i))
            if ent["binary"]:
                rhs[0::2] &= np.uint64(1)
                rhs[1::2] = 0
        else:
            dt = np.uint64 if wb == 64 else np.uint32
            lhs = np.array([L.orc_splitmix64(seed + i) & ((1 << wb) - 1) for i in range(n)], dtype=dt)
            rhs = np.array([L.orc_splitmix64(seed + 7777 + i) & ((1 << wb) - 1) for i in range(n)], dtype=dt)
            if ent["binary"]:
                rhs &= dt(1)
        prod = np.zeros_like(lhs)
        nat.negacyclic_polymul(prod, lhs, rhs)
        assert sha(prod) == ent["prod_sha256"], (kind, n)
        if "prod" in ent and wb != 128:
            assert [str(int(x)) for x in prod] == ent["prod"]
        # the reference's own oracle (schoolbook) agrees too: src/native64.rs:1208-1215
        if not ent["binary"] or True:
            conv = oracle.negacyclic_convolution(n, 0, lhs, rhs, wb)
            assert np.array_equal(conv, prod)


def test_try_new_none_and_panics(oracle, golden):
    for c in golden["try_new_none"]:
        assert oracle.Plan.try_new(c["n"], c["p"], c["bits"]) is None, c
    for p in (0, 1):  # Div64::new / Div32::new panic: src/fastdiv.rs:48-49,98-99
        with pytest.raises(ValueError):
            oracle.Plan.try_new(64, p, 64)
        with pytest.raises(ValueError):
            oracle.Plan.try_new(64, p, 32)


def test_prime_search(oracle, golden):
    L = oracle.lib()
    for c in golden["prime_search"]:  # src/prime.rs:220-221
        out = ctypes.c_uint64(0)
        assert L.orc_largest_prime_in_arithmetic_progression64(c["factor"], c["offset"], c["lo"], c["hi"],
                                                               ctypes.addressof(out))
        assert out.value == c["value"]
    # benches/ntt.rs:84-118 prime grid
    grid = [((1 << 16, 1, 1 << 29, 1 << 30), 1073479681), ((1 << 16, 1, 1 << 30, 1 << 31), 2147352577),
            ((1 << 16, 1, 1 << 31, 1 << 32), 4293918721), ((1 << 16, 1, 1 << 49, 1 << 50), 1125899904679937),
            ((1 << 16, 1, 1 << 50, 1 << 51), 2251799813554177),
            ((1 << 16, 1, 1 << 61, 1 << 62), 4611686018427322369),
            ((1 << 16, 1, 1 << 62, 1 << 63), 9223372036853661697),
            ((1 << 16, 1, 1 << 63, (1 << 64) - 1), 18446744073707716609)]
    for args, val in grid:
        out = ctypes.c_uint64(0)
        assert L.orc_largest_prime_in_arithmetic_progression64(*args, ctypes.addressof(out))
        assert out.value == val


def test_avx512_restatement_equals_scalar_engine(oracle):
    """oracle's AVX-512 restatement of the 62-bit-class engine (src/prime64/shoup.rs:10-156, used for the CPU
    baseline on AVX-512 hosts) against its scalar engine: same words for every size, both directions."""
    import subprocess
    if "avx512dq" not in open("/proc/cpuinfo").read():
        pytest.skip("host CPU has no AVX-512F+DQ")
    oracle.build(native=True)
    for p, sizes in ((4611686018427322369, (16, 32, 64, 128, 1024, 2048, 8192)), (1125899904679937, (16, 512)),
                     (9223372036853661697, (64,))):   # the last one is the 63-bit class: falls back to scalar
        for n in sizes:
            plan = oracle.Plan.try_new(n, p, 64, native=True)
            assert plan.avx512_available()
            a = oracle.fill_uniform(n, p, 3 * n + 1, 64)
            x, y = a.copy(), a.copy()
            plan.fwd(x)
            plan.fwd_avx512(y)
            assert np.array_equal(x, y), ("fwd", n, p)
            plan.inv(x)
            plan.inv_avx512(y)
            assert np.array_equal(x, y), ("inv", n, p)
