"""product::Plan (src/product.rs): the oracle against the committed big-integer golden vectors (CPU), the host
side of the HIP library (try_new parity, CPU), and the HIP path through the C ABI against both (GPU).
The cases are the reference's own tests (src/product.rs:976-1166) plus FwdMode::Bounded, InvMode::Accumulate
on every shape, the pointwise calls, and the batched plane-major layout.  Bit-exact throughout."""
import hashlib
import json
import os

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


@pytest.fixture(scope="module")
def gp():
    with open(os.path.join(ROOT, "tests", "golden", "golden_product_v1.json")) as f:
        return json.load(f)


def _case_inputs(oracle, c):
    n, big = c["n"], int(c["modulus"])
    a = oracle.fill_uniform(n, big, c["seed_a"], 64)
    b = oracle.fill_uniform(n, big, c["seed_b"], 64)
    init = oracle.fill_uniform(n, big, c["seed_init"], 64)
    return a, b, init


def _bounded_input(oracle, c):
    bd, big = c["bounded"], int(c["modulus"])
    raw = oracle.fill_uniform(c["n"], 2 * bd["bound"] - 1, bd["seed"], 64)
    centred = [int(x) - bd["bound"] + 1 for x in raw]
    std = np.array([x % big for x in centred], dtype=np.uint64)
    assert sha(std) == bd["standard_sha256"]
    return std


def _run_case(make_plan, fwd_mode, c, oracle):
    """Every golden field of one case through `make_plan`'s implementation (oracle or HIP)."""
    n, big = c["n"], int(c["modulus"])
    factors = [int(x) for x in c["factors"]]
    plan = make_plan(n, big, factors)
    assert plan is not None
    a, b, init = _case_inputs(oracle, c)
    dl = plan.ntt_domain_len()
    assert dl == (n // 2) * c["n32"] + n * c["n64"]
    fa, fb = np.zeros(dl, dtype=np.uint64), np.zeros(dl, dtype=np.uint64)
    plan.fwd(fa, a, fwd_mode(None))
    plan.fwd(fb, b, fwd_mode(None))
    assert sha(fa) == c["fwd_sha256"]
    if "fwd" in c:
        assert [int(x) for x in fa] == [int(x) for x in c["fwd"]]
    out = np.zeros(n, dtype=np.uint64)
    t = fa.copy()
    plan.inv(out, t, 0)
    assert sha(out) == c["inv_replace_sha256"]
    acc = init.copy()
    t = fa.copy()
    plan.inv(acc, t, 1)
    assert sha(acc) == c["inv_accumulate_sha256"]
    t = fa.copy()
    plan.mul_assign_normalize(t, fb)
    assert sha(t) == c["mul_assign_normalize_sha256"]
    plan.inv(out, t, 0)
    assert sha(out) == c["polymul_sha256"]
    t = fa.copy()
    plan.normalize(t)
    assert sha(t) == c["normalize_sha256"]
    t = fa.copy()
    plan.mul_accumulate(t, fa, fb)
    assert sha(t) == c["mul_accumulate_sha256"]
    if "bounded" in c:
        std = _bounded_input(oracle, c)
        t = np.zeros(dl, dtype=np.uint64)
        plan.fwd(t, std, fwd_mode(c["bounded"]["bound"]))
        assert sha(t) == c["bounded"]["fwd_sha256"]
        g = np.zeros(dl, dtype=np.uint64)
        plan.fwd(g, std, fwd_mode(None))
        assert np.array_equal(g, t)


# ------------------------------------------------------------------------------------------------
# CPU: oracle vs golden; host side of the library vs oracle
# ------------------------------------------------------------------------------------------------
def test_oracle_product_golden(oracle, gp):
    for c in gp["cases"]:
        _run_case(lambda n, m, f: oracle.Product.try_new(n, m, f), lambda b: b, c, oracle)


def test_oracle_product_none(oracle, gp):
    for c in gp["none"]:
        assert oracle.Product.try_new(c["n"], int(c["modulus"]), [int(x) for x in c["factors"]]) is None, c["why"]


def test_host_product_try_new(oracle, gp):
    """Plan::try_new on the host side of libcntt_hip.so: no GPU needed (device tables upload lazily)."""
    from concrete_ntt_amd import product
    for c in gp["none"]:
        assert product.Plan.try_new(c["n"], int(c["modulus"]), [int(x) for x in c["factors"]]) is None, c["why"]
    for c in gp["cases"]:
        f = [int(x) for x in c["factors"]]
        pl = product.Plan.try_new(c["n"], int(c["modulus"]), f)
        opl = oracle.Product.try_new(c["n"], int(c["modulus"]), f)
        assert pl is not None and pl.ntt_size() == c["n"] and pl.modulus() == int(c["modulus"])
        assert pl.ntt_domain_len() == opl.ntt_domain_len()
        assert pl.primes() == sorted(x for x in f if x != 1)
        assert (len(pl.plan_32()), len(pl.plan_64())) == (c["n32"], c["n64"])
        assert np.array_equal(pl.modular_inverses(), opl.modular_inverses())
        for sub, p in zip(pl.plan_32() + pl.plan_64(), pl.primes()):
            assert sub.modulus() == p and sub.ntt_size() == c["n"]
        assert pl.clone().primes() == pl.primes()
    # a plan without primes (modulus 1): the release-build behaviour of src/product.rs:205-206
    pl = product.Plan.try_new(64, 1, [])
    assert pl is not None and pl.ntt_domain_len() == 0


def test_product_modes_api():
    from concrete_ntt_amd import product
    assert repr(product.FwdMode.Generic) == "Generic" and repr(product.FwdMode.Bounded(7)) == "Bounded(7)"
    assert (product.InvMode.Replace, product.InvMode.Accumulate) == (0, 1)


# ------------------------------------------------------------------------------------------------
# GPU: the HIP path through the C ABI
# ------------------------------------------------------------------------------------------------
def _hip_plan(n, m, f):
    from concrete_ntt_amd import product
    return product.Plan.try_new(n, m, f)


def _hip_mode(b):
    from concrete_ntt_amd import product
    return product.FwdMode.Generic if b is None else product.FwdMode.Bounded(b)


@pytest.mark.gpu
def test_gpu_product_golden(oracle, gp):
    for c in gp["cases"]:
        _run_case(_hip_plan, _hip_mode, c, oracle)


def _ref_primes(oracle, n, shape):
    lp = oracle.largest_prime_in_arithmetic_progression64
    f = 2 * n
    if shape == "u64x1":
        return [lp(f, 1, 0, 2**64 - 1)]
    if shape == "u32x1":
        return [lp(f, 1, 0, 2**32 - 1)]
    if shape == "u32x2":
        p0 = lp(f, 1, 0, 2**32 - 1)
        return [p0, lp(f, 1, 0, p0 - 1)]
    if shape == "u30x2":
        p0 = lp(f, 1, 0, 2**30)
        return [p0, lp(f, 1, 0, p0 - 1)]
    if shape == "u32x4":
        ps = [lp(f, 1, 0, 2**16 - 1)]
        for _ in range(3):
            if ps[-1] is None:
                pytest.skip("fewer than four primes = 1 mod 2n below 2^16")
            ps.append(lp(f, 1, 0, ps[-1] - 1))
        if ps[-1] is None:
            pytest.skip("fewer than four primes = 1 mod 2n below 2^16")
        return ps
    if shape == "u32x2_u64x1":
        if n <= 1024:  # the reference's choice, src/product.rs:1125-1127
            p1 = lp(f, 1, 0, 2**15)
            return [lp(f, 1, 0, 2**33), p1, lp(f, 1, 0, p1 - 1)]
        p1 = lp(f, 1, 0, 2**16 + 1)
        return [lp(f, 1, 0, 2**32 + 2**24), p1, lp(f, 1, 0, p1 - 1)]
    if shape == "u64x2":
        p0 = lp(f, 1, 0, 2**32 + 2**24)
        return [p0, lp(f, 1, 0, 2**31)]
    raise KeyError(shape)


SHAPES = ["u64x1", "u32x1", "u32x2", "u30x2", "u32x4", "u32x2_u64x1", "u64x2"]


@pytest.mark.gpu
@pytest.mark.parametrize("shape", SHAPES)
@pytest.mark.parametrize("n", [256, 1024, 4096])
def test_gpu_product_reference_roundtrip(oracle, shape, n):
    """src/product.rs:976-1147: inv(fwd(x)) * n^-1 == x mod p, for Replace and Accumulate (into zeros); the HIP
    buffers are also compared with the oracle's word for word, including the ntt buffer that inv leaves behind."""
    from concrete_ntt_amd import product
    primes = _ref_primes(oracle, n, shape)
    big = 1
    for p in primes:
        big *= p
    plan = product.Plan.try_new(n, big, primes)
    oplan = oracle.Product.try_new(n, big, primes)
    assert plan is not None and oplan is not None
    std = oracle.fill_uniform(n, big, 4242 + n, 64)
    ninv = pow(n, -1, big)
    for mode in (product.InvMode.Replace, product.InvMode.Accumulate):
        ntt = np.zeros(plan.ntt_domain_len(), dtype=np.uint64)
        ontt = ntt.copy()
        plan.fwd(ntt, std, product.FwdMode.Generic)
        oplan.fwd(ontt, std)
        assert np.array_equal(ntt, ontt)
        rt, ort = np.zeros(n, dtype=np.uint64), np.zeros(n, dtype=np.uint64)
        plan.inv(rt, ntt, mode)
        oplan.inv(ort, ontt, mode == product.InvMode.Accumulate)
        assert np.array_equal(rt, ort) and np.array_equal(ntt, ontt)
        assert [int(x) * ninv % big for x in rt] == [int(x) for x in std]


@pytest.mark.gpu
@pytest.mark.parametrize("shape", SHAPES)
def test_gpu_product_batch_layout(oracle, shape):
    """_batch calls on device tensors: standard back to back, NTT domain plane-major; every polynomial equals
    the oracle's single-polynomial result, for fwd, the pointwise calls and inv (both modes)."""
    import torch
    from concrete_ntt_amd import product
    n, batch = 512, 37
    primes = _ref_primes(oracle, n, shape)
    big = 1
    for p in primes:
        big *= p
    plan, oplan = product.Plan.try_new(n, big, primes), oracle.Product.try_new(n, big, primes)
    n32 = sum(p < 2**32 for p in primes)
    n64 = len(primes) - n32
    dl = plan.ntt_domain_len()
    a = oracle.fill_uniform(n * batch, big, 99, 64)
    b = oracle.fill_uniform(n * batch, big, 199, 64)
    init = oracle.fill_uniform(n * batch, big, 299, 64)

    def dev(x):
        return torch.from_numpy(x.view(np.int64).copy()).cuda()

    def host(t):
        return t.cpu().numpy().view(np.uint64)

    def planes_to_poly(buf, i):
        """polynomial i's reference-layout ntt buffer out of a plane-major batch buffer"""
        w32 = buf[: (n // 2) * n32 * batch].view(np.uint32).reshape(n32, batch, n) if n32 else None
        w64 = buf[(n // 2) * n32 * batch:].reshape(n64, batch, n) if n64 else None
        parts = []
        if n32:
            parts.append(np.ascontiguousarray(w32[:, i, :]).reshape(-1).view(np.uint64))
        if n64:
            parts.append(np.ascontiguousarray(w64[:, i, :]).reshape(-1))
        return np.concatenate(parts)

    da, db = dev(a), dev(b)
    fa = torch.zeros(dl * batch, dtype=torch.int64, device="cuda")
    fb = torch.zeros_like(fa)
    plan.fwd_batch(fa, da, product.FwdMode.Generic)
    plan.fwd_batch(fb, db, product.FwdMode.Generic)
    hfa, hfb = host(fa), host(fb)
    ofa, ofb = [], []
    for i in range(batch):
        x, y = np.zeros(dl, dtype=np.uint64), np.zeros(dl, dtype=np.uint64)
        oplan.fwd(x, a[i * n:(i + 1) * n])
        oplan.fwd(y, b[i * n:(i + 1) * n])
        ofa.append(x)
        ofb.append(y)
        assert np.array_equal(planes_to_poly(hfa, i), x), i
    # pointwise on the device buffers
    macc = fa.clone()
    plan.mul_accumulate_batch(macc, fa, fb)
    nrm = fa.clone()
    plan.normalize_batch(nrm)
    prod = fa.clone()
    plan.mul_assign_normalize_batch(prod, fb)
    hm, hn, hp = host(macc), host(nrm), host(prod)
    for i in (0, 1, batch // 2, batch - 1):
        x = ofa[i].copy()
        oplan.mul_accumulate(x, ofa[i], ofb[i])
        assert np.array_equal(planes_to_poly(hm, i), x)
        x = ofa[i].copy()
        oplan.normalize(x)
        assert np.array_equal(planes_to_poly(hn, i), x)
        x = ofa[i].copy()
        oplan.mul_assign_normalize(x, ofb[i])
        assert np.array_equal(planes_to_poly(hp, i), x)
    # inverse, Replace then Accumulate
    out = torch.zeros(n * batch, dtype=torch.int64, device="cuda")
    plan.inv_batch(out, prod.clone(), product.InvMode.Replace)
    acc = dev(init)
    plan.inv_batch(acc, prod, product.InvMode.Accumulate)
    hout, hacc = host(out), host(acc)
    for i in range(batch):
        x = ofa[i].copy()
        oplan.mul_assign_normalize(x, ofb[i])
        r = np.zeros(n, dtype=np.uint64)
        oplan.inv(r, x.copy(), False)
        assert np.array_equal(hout[i * n:(i + 1) * n], r), i
        r2 = init[i * n:(i + 1) * n].copy()
        oplan.inv(r2, x, True)
        assert np.array_equal(hacc[i * n:(i + 1) * n], r2), i
    # the product really is the negacyclic convolution mod p (reference tests' own oracle), polynomial 0
    if big < 2**63:
        want = oracle.negacyclic_convolution(n, big, a[:n].copy(), b[:n].copy(), 64)
        assert np.array_equal(hout[:n], want)


@pytest.mark.gpu
def test_gpu_product_split_edge_values(oracle):
    """The division-free `% p` of the split kernel on boundary inputs (0, p-1, p, 2p, 2^64-1, values around
    multiples of each prime) for every plan shape."""
    from concrete_ntt_amd import product
    n = 256
    for shape in SHAPES:
        primes = _ref_primes(oracle, n, shape)
        if len(primes) == 1:
            continue  # single-prime plans do not reduce (src/product.rs:282-293)
        big = 1
        for p in primes:
            big *= p
        plan, oplan = product.Plan.try_new(n, big, primes), oracle.Product.try_new(n, big, primes)
        vals = [0, 1, 2**64 - 1, 2**63, 2**63 - 1, big - 1, big % 2**64, big // 2, big // 2 + 1]
        for p in primes:
            for m in (1, 2, 3, (2**64 - 1) // p):
                vals += [(m * p + d) % 2**64 for d in (-1, 0, 1)]
        std = np.array((vals * (n // len(vals) + 1))[:n], dtype=np.uint64)
        ntt, ontt = np.zeros(plan.ntt_domain_len(), dtype=np.uint64), np.zeros(plan.ntt_domain_len(), dtype=np.uint64)
        plan.fwd(ntt, std, product.FwdMode.Generic)
        oplan.fwd(ontt, std)
        assert np.array_equal(ntt, ontt), shape


@pytest.mark.gpu
def test_gpu_product_bounded_and_accumulate_wrap(oracle):
    """FwdMode::Bounded on the u32x2 plan (positive and negative centred values; a bound that is too large falls
    back to `%`), and InvMode::Accumulate when standard + result wraps past 2^64 (add_mod_u64's overflow arm)."""
    from concrete_ntt_amd import product
    n = 1024
    primes = _ref_primes(oracle, n, "u32x2")
    big = primes[0] * primes[1]  # close to 2^64, so standard + acc overflows u64 regularly
    plan, oplan = product.Plan.try_new(n, big, primes), oracle.Product.try_new(n, big, primes)
    dl = plan.ntt_domain_len()
    for bound in (1, 2, 1 << 20, min(primes) - 1, min(primes), 1 << 40):
        mag = min(bound, min(primes) - 1)
        raw = oracle.fill_uniform(n, 2 * mag - 1, 5 + bound % 1000, 64)
        std = np.array([(int(x) - mag + 1) % big for x in raw], dtype=np.uint64)
        ntt, ontt = np.zeros(dl, dtype=np.uint64), np.zeros(dl, dtype=np.uint64)
        plan.fwd(ntt, std, product.FwdMode.Bounded(bound))
        oplan.fwd(ontt, std, bound)
        assert np.array_equal(ntt, ontt), bound
    std = oracle.fill_uniform(n, big, 31, 64)
    init = np.full(n, big - 1, dtype=np.uint64)
    ntt, ontt = np.zeros(dl, dtype=np.uint64), np.zeros(dl, dtype=np.uint64)
    plan.fwd(ntt, std, product.FwdMode.Generic)
    oplan.fwd(ontt, std)
    acc, oacc = init.copy(), init.copy()
    plan.inv(acc, ntt, product.InvMode.Accumulate)
    oplan.inv(oacc, ontt, True)
    assert np.array_equal(acc, oacc)
    assert [int(x) for x in acc] == [(big - 1 + int(s) * n) % big for s in std]


@pytest.mark.gpu
def test_gpu_product_length_panics(oracle):
    """assert_eq! on the slice lengths: src/product.rs:275-276, :362-363, :887-888, :919, :937-938."""
    from concrete_ntt_amd import Panic, product
    n = 256
    primes = _ref_primes(oracle, n, "u32x2_u64x1")
    big = primes[0] * primes[1] * primes[2]
    plan = product.Plan.try_new(n, big, primes)
    dl = plan.ntt_domain_len()
    good_s, good_n = np.zeros(n, dtype=np.uint64), np.zeros(dl, dtype=np.uint64)
    with pytest.raises(Panic):
        plan.fwd(good_n, np.zeros(n - 1, dtype=np.uint64))
    with pytest.raises(Panic):
        plan.fwd(np.zeros(dl + 1, dtype=np.uint64), good_s)
    with pytest.raises(Panic):
        plan.inv(good_s, np.zeros(dl - 1, dtype=np.uint64))
    with pytest.raises(Panic):
        plan.mul_assign_normalize(good_n, np.zeros(dl + 2, dtype=np.uint64))
    with pytest.raises(Panic):
        plan.normalize(np.zeros(1, dtype=np.uint64))
    with pytest.raises(Panic):
        plan.mul_accumulate(good_n, np.zeros(dl - 1, dtype=np.uint64), good_n)
    # empty plan: inv Replace zero-fills (src/product.rs:378-384)
    empty = product.Plan.try_new(64, 1, [])
    s = np.arange(64, dtype=np.uint64)
    empty.inv(s, np.zeros(0, dtype=np.uint64), product.InvMode.Replace)
    assert not s.any()


@pytest.mark.gpu
@pytest.mark.parametrize("shape", ["u32x2", "u64x1", "u32x2_u64x1", "u30x2"])
@pytest.mark.parametrize("accumulate", [False, True])
def test_gpu_product_external_product(oracle, shape, accumulate):
    """cntt_product_external_product_batch against the oracle's Plan::fwd / mul_accumulate / inv in sequence
    (src/product.rs:273, :935, :360), Generic and Bounded forward modes, Replace and Accumulate inverse modes."""
    import torch
    from concrete_ntt_amd import product
    n, J, O, batch = 512, 3, 2, 3
    primes = sorted(_ref_primes(oracle, n, shape))
    big = 1
    for p in primes:
        big *= p
    plan, oplan = product.Plan.try_new(n, big, primes), oracle.Product.try_new(n, big, primes)
    n32 = sum(p < 2**32 for p in primes)
    dl = plan.ntt_domain_len()
    bound = 1 << 18
    raw = oracle.fill_uniform(batch * J * n, 2 * bound - 1, 7, 64)
    terms = np.array([(int(x) - bound + 1) % big for x in raw], dtype=np.uint64)  # centred, |value| < bound
    init = oracle.fill_uniform(batch * O * n, big, 8, 64)
    # key: canonical residues per prime, plane-major for a "batch" of J*O polynomials
    planes = [oracle.fill_uniform(J * O * n, p, 50 + i, 64) for i, p in enumerate(primes)]
    key32 = np.concatenate([pl_.astype(np.uint32) for pl_ in planes[:n32]]) if n32 else np.zeros(0, dtype=np.uint32)
    key = np.concatenate([key32.view(np.uint64)] + [pl_ for pl_ in planes[n32:]])

    def key_poly(j, o):  # reference-layout ntt buffer of key[j][o]
        i = j * O + o
        parts = []
        if n32:
            parts.append(np.concatenate([planes[k][i * n:(i + 1) * n].astype(np.uint32) for k in range(n32)]).view(np.uint64))
        parts += [planes[k][i * n:(i + 1) * n] for k in range(n32, len(primes))]
        return np.concatenate(parts)

    for fwd_bound in (None, bound):
        want = np.zeros(batch * O * n, dtype=np.uint64)
        for b in range(batch):
            acc = [np.zeros(dl, dtype=np.uint64) for _ in range(O)]
            for j in range(J):
                t = np.zeros(dl, dtype=np.uint64)
                oplan.fwd(t, terms[(b * J + j) * n:(b * J + j + 1) * n].copy(), fwd_bound)
                for o in range(O):
                    oplan.mul_accumulate(acc[o], t, key_poly(j, o))
            for o in range(O):
                r = init[(b * O + o) * n:(b * O + o + 1) * n].copy() if accumulate else np.zeros(n, dtype=np.uint64)
                oplan.inv(r, acc[o], accumulate)
                want[(b * O + o) * n:(b * O + o + 1) * n] = r
        mode = product.FwdMode.Generic if fwd_bound is None else product.FwdMode.Bounded(fwd_bound)
        imode = product.InvMode.Accumulate if accumulate else product.InvMode.Replace
        dout = torch.from_numpy((init if accumulate else np.zeros_like(init)).view(np.int64).copy()).cuda()
        dterms = torch.from_numpy(terms.view(np.int64).copy()).cuda()
        dkey = torch.from_numpy(key.view(np.int64).copy()).cuda()
        plan.external_product_batch(dout, dterms, dkey, J, O, mode, imode)
        assert np.array_equal(dout.cpu().numpy().view(np.uint64), want), (shape, fwd_bound)
        hout = (init if accumulate else np.zeros_like(init)).copy()
        plan.external_product_batch(hout, terms, key, J, O, mode, imode)   # host-memory call
        assert np.array_equal(hout, want)
