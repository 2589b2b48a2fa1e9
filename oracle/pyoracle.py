"""ctypes front-end of the CPU oracle (TEST INFRASTRUCTURE ONLY, NOT THE PRODUCT).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
The shipped package (concrete-ntt_amd/) never does.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIBS = {}

c_sz, c_u32, c_u64, c_vp, c_int, c_dbl = (ctypes.c_size_t, ctypes.c_uint32, ctypes.c_uint64,
                                           ctypes.c_void_p, ctypes.c_int, ctypes.c_double)

NATIVE_KINDS = {
    "native32_plan32": 0, "native64_plan32": 1, "native128_plan32": 2,
    "native_binary32_plan32": 3, "native_binary64_plan32": 4, "native_binary128_plan32": 5,
    "native32_plan52": 6, "native64_plan52": 7, "native_binary32_plan52": 8, "native_binary64_plan52": 9,
}
NATIVE_INFO = {  # kind: (nprimes, word bytes, is52, binary)
    "native32_plan32": (3, 4, False, False), "native64_plan32": (5, 8, False, False),
    "native128_plan32": (10, 16, False, False), "native_binary32_plan32": (2, 4, False, True),
    "native_binary64_plan32": (3, 8, False, True), "native_binary128_plan32": (5, 16, False, True),
    "native32_plan52": (2, 4, True, False), "native64_plan52": (3, 8, True, False),
    "native_binary32_plan52": (1, 4, True, True), "native_binary64_plan52": (2, 8, True, True),
}


def build(native=False):
    """Compile the C restatement (building the checker is not using it)."""
    target = "native" if native else "all"
    # several processes may arrive here at once (the ranks of `bench.py --gpus N` verify their results against this checker): one builds,
    # the others wait for it
    import fcntl
    os.makedirs(os.path.join(_HERE, "_build"), exist_ok=True)
    with open(os.path.join(_HERE, "_build", ".build.lock"), "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        try:
            subprocess.run(["make", "-s", "-C", _HERE, target], check=True)
        finally:
            fcntl.flock(lock, fcntl.LOCK_UN)
    return os.path.join(_HERE, "_build", "libcntt_oracle_native.so" if native else "libcntt_oracle.so")


def lib(native=False):
    key = bool(native)
    if key in _LIBS:
        return _LIBS[key]
    path = os.path.join(_HERE, "_build", "libcntt_oracle_native.so" if native else "libcntt_oracle.so")
    src_m = max(os.path.getmtime(os.path.join(_HERE, f)) for f in ("cntt_oracle.c", "cntt_oracle.h"))
    if not os.path.exists(path) or os.path.getmtime(path) < src_m:
        build(native)
    L = ctypes.CDLL(path)
    sig = {
        "orc_bit_rev": (c_sz, [c_u32, c_sz]),
        "orc_mul_mod64": (c_u64, [c_u64, c_u64, c_u64]),
        "orc_exp_mod64": (c_u64, [c_u64, c_u64, c_u64]),
        "orc_is_prime64": (c_int, [c_u64]),
        "orc_largest_prime_in_arithmetic_progression64": (c_int, [c_u64, c_u64, c_u64, c_u64, c_vp]),
        "orc_get_z64": (c_int, [c_u64, c_vp]),
        "orc_find_primitive_root64": (c_int, [c_u64, c_u64, c_vp]),
        "orc_plan64_try_new": (c_vp, [c_sz, c_u64, c_vp]),
        "orc_plan64_free": (None, [c_vp]),
        "orc_plan64_fwd": (None, [c_vp, c_vp]),
        "orc_plan64_inv": (None, [c_vp, c_vp]),
        "orc_plan64_mul_assign_normalize": (None, [c_vp, c_vp, c_vp, c_sz]),
        "orc_plan64_normalize": (None, [c_vp, c_vp, c_sz]),
        "orc_plan64_mul_accumulate": (None, [c_vp, c_vp, c_vp, c_vp, c_sz]),
        "orc_plan32_try_new": (c_vp, [c_sz, c_u32, c_vp]),
        "orc_plan32_free": (None, [c_vp]),
        "orc_plan32_fwd": (None, [c_vp, c_vp]),
        "orc_plan32_inv": (None, [c_vp, c_vp]),
        "orc_plan32_mul_assign_normalize": (None, [c_vp, c_vp, c_vp, c_sz]),
        "orc_plan32_normalize": (None, [c_vp, c_vp, c_sz]),
        "orc_plan32_mul_accumulate": (None, [c_vp, c_vp, c_vp, c_vp, c_sz]),
        "orc_primes32_p": (c_u32, [c_int]),
        "orc_primes52_p": (c_u64, [c_int]),
        "orc_reconstruct_32bit_01": (c_u32, [c_u32, c_u32]),
        "orc_reconstruct_32bit_012_u32": (c_u32, [c_u32, c_u32, c_u32]),
        "orc_reconstruct_32bit_012_u64": (c_u64, [c_u32, c_u32, c_u32]),
        "orc_reconstruct_32bit_01234_v2_u64": (c_u64, [c_vp]),
        "orc_reconstruct_52bit_0": (c_u32, [c_u64]),
        "orc_reconstruct_52bit_01_u32": (c_u32, [c_u64, c_u64]),
        "orc_reconstruct_52bit_01_u64": (c_u64, [c_u64, c_u64]),
        "orc_reconstruct_52bit_012": (c_u64, [c_u64, c_u64, c_u64]),
        "orc_native_try_new": (c_vp, [c_int, c_sz]),
        "orc_native_free": (None, [c_vp]),
        "orc_native_fwd": (None, [c_vp, c_vp, c_vp]),
        "orc_native_fwd_binary": (None, [c_vp, c_vp, c_vp]),
        "orc_native_inv": (None, [c_vp, c_vp, c_vp]),
        "orc_native_negacyclic_polymul": (None, [c_vp, c_vp, c_vp, c_vp]),
        "orc_product_try_new": (c_vp, [c_sz, c_u64, c_vp, c_sz]),
        "orc_product_free": (None, [c_vp]),
        "orc_product_ntt_domain_len": (c_sz, [c_vp]),
        "orc_product_modular_inverses": (c_sz, [c_vp, c_vp]),
        "orc_product_fwd": (None, [c_vp, c_vp, c_vp, c_int, c_u64]),
        "orc_product_inv": (None, [c_vp, c_vp, c_vp, c_int]),
        "orc_product_mul_assign_normalize": (None, [c_vp, c_vp, c_vp]),
        "orc_product_normalize": (None, [c_vp, c_vp]),
        "orc_product_mul_accumulate": (None, [c_vp, c_vp, c_vp, c_vp]),
        "orc_negacyclic_convolution64": (None, [c_sz, c_u64, c_vp, c_vp, c_vp]),
        "orc_negacyclic_convolution32": (None, [c_sz, c_u32, c_vp, c_vp, c_vp]),
        "orc_negacyclic_convolution128": (None, [c_sz, c_vp, c_vp, c_vp]),
        "orc_splitmix64": (c_u64, [c_u64]),
        "orc_fill_uniform_u64": (None, [c_vp, c_sz, c_u64, c_u64]),
        "orc_fill_uniform_u32": (None, [c_vp, c_sz, c_u32, c_u64]),
        "orc_plan64_fwd_batch": (c_dbl, [c_vp, c_vp, c_sz, c_int]),
        "orc_plan64_inv_batch": (c_dbl, [c_vp, c_vp, c_sz, c_int]),
        "orc_plan64_mul_assign_normalize_batch": (c_dbl, [c_vp, c_vp, c_vp, c_sz, c_int]),
        "orc_avx512_available": (c_int, []),
        "orc_plan64_fwd_avx512": (None, [c_vp, c_vp]),
        "orc_plan64_inv_avx512": (None, [c_vp, c_vp]),
        "orc_plan64_fwd_avx512_batch": (c_dbl, [c_vp, c_vp, c_sz, c_int]),
        "orc_plan64_inv_avx512_batch": (c_dbl, [c_vp, c_vp, c_sz, c_int]),
        "orc_plan64_fwd_inv_loop": (c_dbl, [c_vp, c_vp, c_sz, c_int, c_int, c_int]),
        "orc_plan32_fwd_batch": (c_dbl, [c_vp, c_vp, c_sz, c_int]),
        "orc_plan32_inv_batch": (c_dbl, [c_vp, c_vp, c_sz, c_int]),
        "orc_native_negacyclic_polymul_batch": (c_dbl, [c_vp, c_vp, c_vp, c_vp, c_sz, c_int]),
    }
    for name, (res, args) in sig.items():
        fn = getattr(L, name)
        fn.restype, fn.argtypes = res, args
    _LIBS[key] = L
    return L


def _ptr(a):
    return a.ctypes.data


class _PlanStruct64(ctypes.Structure):
    _fields_ = [("n", c_sz), ("p", c_u64), ("twid", c_vp), ("twid_shoup", c_vp), ("inv_twid", c_vp),
                ("inv_twid_shoup", c_vp), ("p_barrett", c_u64), ("big_q", c_u64), ("n_inv_mod_p", c_u64),
                ("n_inv_mod_p_shoup", c_u64)]


class _PlanStruct32(ctypes.Structure):
    _fields_ = [("n", c_sz), ("p", c_u32), ("twid", c_vp), ("twid_shoup", c_vp), ("inv_twid", c_vp),
                ("inv_twid_shoup", c_vp), ("p_barrett", c_u32), ("big_q", c_u32), ("n_inv_mod_p", c_u32),
                ("n_inv_mod_p_shoup", c_u32)]


class Plan:
    """prime32::Plan / prime64::Plan of the oracle (bits = 32 or 64)."""

    def __init__(self, handle, bits, native=False):
        self._h, self.bits, self._L = handle, bits, lib(native)
        self.dtype = np.uint64 if bits == 64 else np.uint32
        st = (_PlanStruct64 if bits == 64 else _PlanStruct32).from_address(handle)
        self.n, self.p = st.n, st.p
        self.p_barrett, self.big_q = st.p_barrett, st.big_q
        self.n_inv_mod_p, self.n_inv_mod_p_shoup = st.n_inv_mod_p, st.n_inv_mod_p_shoup
        self._st = st

    @classmethod
    def try_new(cls, n, p, bits, native=False):
        L = lib(native)
        panicked = c_int(0)
        fn = L.orc_plan64_try_new if bits == 64 else L.orc_plan32_try_new
        h = fn(n, p, ctypes.addressof(panicked))
        if panicked.value:
            raise ValueError("reference panics: divisor <= 1 (src/fastdiv.rs:48,99)")
        return cls(h, bits, native) if h else None

    def _f(self, name):
        return getattr(self._L, "orc_plan%d_%s" % (self.bits, name))

    def table(self, name):
        ptr = getattr(self._st, name)
        if not ptr:
            return None
        ct = (c_u64 if self.bits == 64 else c_u32) * self.n
        return np.frombuffer(ct.from_address(ptr), dtype=self.dtype).copy()

    def fwd(self, buf):
        assert buf.dtype == self.dtype and buf.size == self.n and buf.flags.c_contiguous
        self._f("fwd")(self._h, _ptr(buf))

    def inv(self, buf):
        assert buf.dtype == self.dtype and buf.size == self.n and buf.flags.c_contiguous
        self._f("inv")(self._h, _ptr(buf))

    def mul_assign_normalize(self, lhs, rhs):
        self._f("mul_assign_normalize")(self._h, _ptr(lhs), _ptr(rhs), min(lhs.size, rhs.size))

    def normalize(self, values):
        self._f("normalize")(self._h, _ptr(values), values.size)

    def mul_accumulate(self, acc, lhs, rhs):
        self._f("mul_accumulate")(self._h, _ptr(acc), _ptr(lhs), _ptr(rhs), min(acc.size, lhs.size, rhs.size))

    def avx512_available(self):
        """True when this build carries the AVX-512 restatement (62-bit class, u64) and the CPU runs it."""
        return self.bits == 64 and bool(self._L.orc_avx512_available())

    def fwd_avx512(self, buf):
        assert self.bits == 64 and buf.dtype == np.uint64 and buf.size == self.n
        self._L.orc_plan64_fwd_avx512(self._h, _ptr(buf))

    def inv_avx512(self, buf):
        assert self.bits == 64 and buf.dtype == np.uint64 and buf.size == self.n
        self._L.orc_plan64_inv_avx512(self._h, _ptr(buf))

    def fwd_avx512_batch(self, bufs, nthreads=1):
        return self._L.orc_plan64_fwd_avx512_batch(self._h, _ptr(bufs), bufs.size // self.n, nthreads)

    def inv_avx512_batch(self, bufs, nthreads=1):
        return self._L.orc_plan64_inv_avx512_batch(self._h, _ptr(bufs), bufs.size // self.n, nthreads)

    def fwd_inv_loop(self, bufs, nthreads=1, reps=1, avx512=False):
        """seconds for `reps` x (fwd, inv) of every polynomial in bufs, split over nthreads (CPU-baseline driver)."""
        return self._L.orc_plan64_fwd_inv_loop(self._h, _ptr(bufs), bufs.size // self.n, nthreads, reps, 1 if avx512 else 0)

    def fwd_batch(self, bufs, nthreads=1):
        assert bufs.dtype == self.dtype and bufs.size % self.n == 0
        return self._f("fwd_batch")(self._h, _ptr(bufs), bufs.size // self.n, nthreads)

    def inv_batch(self, bufs, nthreads=1):
        assert bufs.dtype == self.dtype and bufs.size % self.n == 0
        return self._f("inv_batch")(self._h, _ptr(bufs), bufs.size // self.n, nthreads)

    def mul_assign_normalize_batch(self, lhs, rhs, nthreads=1):
        assert self.bits == 64
        return self._L.orc_plan64_mul_assign_normalize_batch(self._h, _ptr(lhs), _ptr(rhs), lhs.size // self.n,
                                                             nthreads)

    def __del__(self):
        try:
            self._f("free")(self._h)
        except Exception:
            pass


def word_dtype(word):
    """numpy layout of one coefficient: u32, u64, or (lo, hi) u64 pairs for u128."""
    return {4: np.uint32, 8: np.uint64, 16: np.uint64}[word]


class Native:
    def __init__(self, kind, n, native=False):
        self._L = lib(native)
        self.kind, self.n = kind, n
        self.nprimes, self.word, self.is52, self.binary = NATIVE_INFO[kind]
        self.res_dtype = np.uint64 if self.is52 else np.uint32
        self._h = self._L.orc_native_try_new(NATIVE_KINDS[kind], n)
        if not self._h:
            raise ValueError("None")

    @classmethod
    def try_new(cls, kind, n, native=False):
        try:
            return cls(kind, n, native)
        except ValueError:
            return None

    def words(self, count=None):
        n = self.n if count is None else count
        return np.zeros(n * (2 if self.word == 16 else 1), dtype=word_dtype(self.word))

    def residues(self):
        return [np.zeros(self.n, dtype=self.res_dtype) for _ in range(self.nprimes)]

    def _pp(self, res):
        arr = (c_vp * self.nprimes)(*[_ptr(r) for r in res])
        return arr

    def fwd(self, value, res):
        self._L.orc_native_fwd(self._h, _ptr(value), self._pp(res))

    def fwd_binary(self, value, res):
        assert self.binary
        self._L.orc_native_fwd_binary(self._h, _ptr(value), self._pp(res))

    def inv(self, value, res):
        self._L.orc_native_inv(self._h, _ptr(value), self._pp(res))

    def negacyclic_polymul(self, prod, lhs, rhs):
        self._L.orc_native_negacyclic_polymul(self._h, _ptr(prod), _ptr(lhs), _ptr(rhs))

    def negacyclic_polymul_batch(self, prod, lhs, rhs, batch, nthreads=1):
        return self._L.orc_native_negacyclic_polymul_batch(self._h, _ptr(prod), _ptr(lhs), _ptr(rhs), batch,
                                                           nthreads)

    def __del__(self):
        try:
            self._L.orc_native_free(self._h)
        except Exception:
            pass


class Product:
    """product::Plan (src/product.rs:139-967). fwd mode: bound=None -> FwdMode::Generic, else
    FwdMode::Bounded(bound); inv mode: accumulate=False -> InvMode::Replace."""

    def __init__(self, handle, n):
        self._L = lib()
        self._h = handle
        self.n = n

    @classmethod
    def try_new(cls, n, modulus, factors):
        f = np.ascontiguousarray(np.array(list(factors), dtype=np.uint64))
        h = lib().orc_product_try_new(n, modulus, _ptr(f), f.size)
        return cls(h, n) if h else None

    def ntt_domain_len(self):
        return self._L.orc_product_ntt_domain_len(self._h)

    def modular_inverses(self):
        out = np.zeros(120, dtype=np.uint64)
        return out[: self._L.orc_product_modular_inverses(self._h, _ptr(out))].copy()

    def fwd(self, ntt, standard, bound=None):
        assert ntt.dtype == np.uint64 and ntt.size == self.ntt_domain_len() and standard.size == self.n
        self._L.orc_product_fwd(self._h, _ptr(ntt), _ptr(standard), 0 if bound is None else 1, 0 if bound is None else bound)

    def inv(self, standard, ntt, accumulate=False):
        assert ntt.size == self.ntt_domain_len() and standard.size == self.n
        self._L.orc_product_inv(self._h, _ptr(standard), _ptr(ntt), 1 if accumulate else 0)

    def mul_assign_normalize(self, lhs, rhs):
        self._L.orc_product_mul_assign_normalize(self._h, _ptr(lhs), _ptr(rhs))

    def normalize(self, values):
        self._L.orc_product_normalize(self._h, _ptr(values))

    def mul_accumulate(self, acc, lhs, rhs):
        self._L.orc_product_mul_accumulate(self._h, _ptr(acc), _ptr(lhs), _ptr(rhs))

    def __del__(self):
        try:
            self._L.orc_product_free(self._h)
        except Exception:
            pass


def largest_prime_in_arithmetic_progression64(factor, offset, lo, hi):
    out = ctypes.c_uint64(0)
    ok = lib().orc_largest_prime_in_arithmetic_progression64(factor, offset, lo, hi, ctypes.byref(out))
    return out.value if ok else None


def fill_uniform(count, bound, seed, bits):
    L = lib()
    out = np.empty(count, dtype=np.uint64 if bits == 64 else np.uint32)
    (L.orc_fill_uniform_u64 if bits == 64 else L.orc_fill_uniform_u32)(_ptr(out), count, bound, seed)
    return out


def negacyclic_convolution(n, p, lhs, rhs, bits):
    L = lib()
    out = np.zeros_like(lhs)
    if bits == 128:
        L.orc_negacyclic_convolution128(n, _ptr(lhs), _ptr(rhs), _ptr(out))
    elif bits == 64:
        L.orc_negacyclic_convolution64(n, p, _ptr(lhs), _ptr(rhs), _ptr(out))
    else:
        L.orc_negacyclic_convolution32(n, p, _ptr(lhs), _ptr(rhs), _ptr(out))
    return out
