"""concrete_ntt::product -- negacyclic NTT modulo a product of distinct primes (src/product.rs).

    plan = product.Plan.try_new(n, p0 * p1, [p0, p1])
    ntt = numpy.zeros(plan.ntt_domain_len(), dtype=numpy.uint64)
    plan.fwd(ntt, standard, product.FwdMode.Generic)
    plan.inv(standard, ntt, product.InvMode.Replace)
"""
import ctypes

import numpy as np

from . import _lib
from ._lib import Panic, buffer_info, check, lib
from .prime32 import Plan as _Plan32
from .prime64 import Plan as _Plan64


class FwdMode:
    """enum FwdMode { Generic, Bounded(u64) }  (src/product.rs:124-129)"""

    def __init__(self, bound=None):
        self.bound = bound

    @classmethod
    def Bounded(cls, bound):
        return cls(int(bound))

    def __repr__(self):
        return "Generic" if self.bound is None else "Bounded(%d)" % self.bound


FwdMode.Generic = FwdMode()


class InvMode:
    """enum InvMode { Replace, Accumulate }  (src/product.rs:131-136)"""
    Replace = 0
    Accumulate = 1


class Plan:
    """product::Plan (src/product.rs:139-149, :151-967)."""

    def __init__(self, handle):
        self._h = handle
        L = lib()
        self._n = L.cntt_product_ntt_size(handle)
        self._len = L.cntt_product_ntt_domain_len(handle)

    @classmethod
    def try_new(cls, polynomial_size, modulus, factors):
        """Plan::try_new (src/product.rs:153-247) -> plan or None."""
        f = np.ascontiguousarray(np.array([int(x) for x in factors], dtype=np.uint64))
        out = ctypes.c_void_p()
        rc = lib().cntt_product_plan_new(polynomial_size, modulus, f.ctypes.data, f.size, ctypes.byref(out))
        if rc == _lib.NONE:
            return None
        check(rc)
        return cls(out.value)

    def clone(self):
        return Plan(lib().cntt_product_plan_clone(self._h))

    def __del__(self):
        try:
            if self._h:
                lib().cntt_product_plan_free(self._h)
        except Exception:
            pass

    def __repr__(self):
        return "Plan { polynomial_size: %d, modulus: %d }" % (self._n, self.modulus())

    def ntt_size(self):
        return self._n

    def modulus(self):
        return lib().cntt_product_modulus(self._h)

    def ntt_domain_len(self):
        return self._len

    # private fields, for parity tests
    def primes(self):
        L = lib()
        k = L.cntt_product_nprimes32(self._h) + L.cntt_product_nprimes64(self._h)
        return [L.cntt_product_prime(self._h, i) for i in range(k)]

    def plan_32(self):
        L = lib()
        return [_Plan32(L.cntt_product_ntt32(self._h, i), owned=False, parent=self) for i in range(L.cntt_product_nprimes32(self._h))]

    def plan_64(self):
        L = lib()
        return [_Plan64(L.cntt_product_ntt64(self._h, i), owned=False, parent=self) for i in range(L.cntt_product_nprimes64(self._h))]

    def modular_inverses(self):
        k = len(self.primes())
        out = np.zeros(k * (k - 1) // 2 if k else 0, dtype=np.uint64)
        check(lib().cntt_product_modular_inverses(self._h, out.ctypes.data, out.size))
        return out

    @staticmethod
    def _arg(buf):
        ptr, count, esz, where, stream = buffer_info(buf)
        if esz != 8:
            raise TypeError("product::Plan buffers are u64 words")
        return ptr, count, where, stream

    @staticmethod
    def _mode(mode):
        if not isinstance(mode, FwdMode):
            raise TypeError("mode must be FwdMode.Generic or FwdMode.Bounded(bound)")
        return (0, 0) if mode.bound is None else (1, mode.bound)

    # -- the reference's slice API (host memory, one polynomial) -------------------------------
    def fwd(self, ntt, standard, mode=FwdMode.Generic):
        np_, nc, nw, _ = self._arg(ntt)
        sp, sc, sw, _ = self._arg(standard)
        if nw != _lib.MEM_HOST or sw != _lib.MEM_HOST:
            raise TypeError("fwd() takes host slices; use fwd_batch() for device tensors")
        m, bound = self._mode(mode)
        check(lib().cntt_product_fwd(self._h, np_, nc, sp, sc, m, bound))

    def inv(self, standard, ntt, mode=InvMode.Replace):
        sp, sc, sw, _ = self._arg(standard)
        np_, nc, nw, _ = self._arg(ntt)
        if nw != _lib.MEM_HOST or sw != _lib.MEM_HOST:
            raise TypeError("inv() takes host slices; use inv_batch() for device tensors")
        check(lib().cntt_product_inv(self._h, sp, sc, np_, nc, int(mode)))

    def mul_assign_normalize(self, lhs, rhs):
        lp, lc, _, _ = self._arg(lhs)
        rp, rc_, _, _ = self._arg(rhs)
        check(lib().cntt_product_mul_assign_normalize(self._h, lp, lc, rp, rc_))

    def normalize(self, values):
        vp, vc, _, _ = self._arg(values)
        check(lib().cntt_product_normalize(self._h, vp, vc))

    def mul_accumulate(self, acc, lhs, rhs):
        ap, ac, _, _ = self._arg(acc)
        lp, lc, _, _ = self._arg(lhs)
        rp, rc_, _, _ = self._arg(rhs)
        check(lib().cntt_product_mul_accumulate(self._h, ap, ac, lp, lc, rp, rc_))

    # -- batched API (include/cntt.h: standard back to back, NTT domain plane-major) ------------
    def _std(self, buf):
        ptr, count, where, stream = self._arg(buf)
        if count % self._n:
            raise Panic("standard length %d is not a multiple of ntt_size %d" % (count, self._n))
        return ptr, count // self._n, where, stream

    def _dom(self, buf, batch=None, where=None):
        ptr, count, w, stream = self._arg(buf)
        if self._len == 0:
            if count:
                raise Panic("ntt buffer must be empty for a plan without primes")
            return ptr, 0 if batch is None else batch, w, stream
        if count % self._len:
            raise Panic("ntt length %d is not a multiple of ntt_domain_len %d" % (count, self._len))
        b = count // self._len
        if (batch is not None and b != batch) or (where is not None and w != where):
            raise Panic("buffers must hold the same number of polynomials and live in the same memory")
        return ptr, b, w, stream

    def fwd_batch(self, ntt, standard, mode=FwdMode.Generic):
        sp, batch, where, stream = self._std(standard)
        np_, _, _, _ = self._dom(ntt, batch, where)
        m, bound = self._mode(mode)
        check(lib().cntt_product_fwd_batch(self._h, np_, sp, batch, m, bound, where, stream))

    def inv_batch(self, standard, ntt, mode=InvMode.Replace):
        sp, batch, where, stream = self._std(standard)
        np_, _, _, _ = self._dom(ntt, batch, where)
        check(lib().cntt_product_inv_batch(self._h, sp, np_, batch, int(mode), where, stream))

    def mul_assign_normalize_batch(self, lhs, rhs):
        lp, batch, where, stream = self._dom(lhs)
        rp, _, _, _ = self._dom(rhs, batch, where)
        check(lib().cntt_product_mul_assign_normalize_batch(self._h, lp, rp, batch, where, stream))

    def normalize_batch(self, values):
        vp, batch, where, stream = self._dom(values)
        check(lib().cntt_product_normalize_batch(self._h, vp, batch, where, stream))

    def mul_accumulate_batch(self, acc, lhs, rhs):
        ap, batch, where, stream = self._dom(acc)
        lp, _, _, _ = self._dom(lhs, batch, where)
        rp, _, _, _ = self._dom(rhs, batch, where)
        check(lib().cntt_product_mul_accumulate_batch(self._h, ap, lp, rp, batch, where, stream))

    def external_product_batch(self, out, terms, key_ntt, nterms, nout, fwd_mode=FwdMode.Generic, inv_mode=InvMode.Replace):
        """out[b][o] (+)= inv(sum_j fwd(terms[b][j]) . key_ntt[j][o]): the values of Plan.fwd / mul_accumulate / inv
        (src/product.rs:273, :935, :360) called in sequence -- split, one fused chain per prime, Garner."""
        op, ob, where, stream = self._std(out)
        tp, tb, tw, _ = self._std(terms)
        if nout <= 0 or ob % nout or tb != (ob // nout) * nterms or tw != where:
            raise Panic("out: batch*nout, terms: batch*nterms polynomials in the same memory")
        kp, _, _, _ = self._dom(key_ntt, nterms * nout, where)
        m, bound = self._mode(fwd_mode)
        check(lib().cntt_product_external_product_batch(self._h, op, tp, kp, nterms, nout, ob // nout, m, bound,
                                                        int(inv_mode), where, stream))
