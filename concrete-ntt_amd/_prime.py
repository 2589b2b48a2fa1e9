"""prime32::Plan / prime64::Plan over the C ABI (shared implementation)."""
import ctypes

import numpy as np

from . import _lib
from ._lib import Panic, buffer_info, check, lib


class PrimePlan:
    """Negacyclic NTT plan for a prime modulus; mirrors concrete_ntt::prime{32,64}::Plan
    (src/prime64.rs:221-236, :701-1129 ; src/prime32.rs:601-616, :627-927).

    INPUT CONTRACT (include/cntt.h): every coefficient handed to fwd / inv / the pointwise calls is canonical, 0 <= x < modulus --
    what the reference's own tests feed and the only range on which its back ends agree with each other.  The
    kernels rely on it (the lazy classes skip the first stage's conditional subtraction); words >= modulus are NOT rejected and
    NOT reduced: the call completes and the affected outputs are unspecified residues (tests/test_gpu_parity.py pins exactly
    that: no fault, canonical inputs of the same batch unaffected).  `check_canonical(buf)` below validates a host array."""

    def check_canonical(self, buf):
        """Raise Panic if a host array holds a word >= modulus (the transforms do not check: see the class docstring)."""
        a = np.asarray(buf)
        if a.dtype.kind == "i":   # device tensors travel as int64 / int32: compare the words as the unsigned words they are
            a = a.view(np.dtype("u%d" % a.dtype.itemsize))
        if a.size and int(a.max()) >= self.modulus():
            raise Panic("coefficient %d >= modulus %d: outside the transforms' input contract" % (int(a.max()), self.modulus()))

    BITS = 64

    def __init__(self, handle, owned=True, parent=None):
        # a borrowed sub-plan (native ntt_i(), product plan_32()/plan_64()) keeps its parent alive: the C++ object
        # belongs to the parent and is freed with it
        self._h, self._owned, self._parent = handle, owned, parent
        self._p = "cntt_prime%d_" % self.BITS
        self._n = getattr(lib(), self._p + "ntt_size")(handle)

    # -- construction ------------------------------------------------------------------------
    @classmethod
    def try_new(cls, polynomial_size, modulus):
        """Plan::try_new -> plan or None; raises Panic where the reference panics (modulus <= 1)."""
        out = ctypes.c_void_p()
        rc = getattr(lib(), "cntt_prime%d_plan_new" % cls.BITS)(polynomial_size, modulus, ctypes.byref(out))
        if rc == _lib.NONE:
            return None
        check(rc)
        return cls(out.value)

    def clone(self):
        return type(self)(getattr(lib(), self._p + "plan_clone")(self._h))

    def __del__(self):
        try:
            if self._owned and self._h:
                getattr(lib(), self._p + "plan_free")(self._h)
        except Exception:
            pass

    def __repr__(self):  # Debug prints only ntt_size and modulus: src/prime64.rs:238-245
        return "Plan { ntt_size: %d, modulus: %d }" % (self.ntt_size(), self.modulus())

    # -- accessors ---------------------------------------------------------------------------
    def ntt_size(self):
        return self._n

    def modulus(self):
        return getattr(lib(), self._p + "modulus")(self._h)

    def info(self):
        out = _lib.PlanInfo()
        check(getattr(lib(), self._p + "plan_info")(self._h, ctypes.byref(out)))
        return out

    def table(self, which):
        out = np.zeros(self._n, dtype=self.dtype)
        rc = getattr(lib(), self._p + "plan_table")(self._h, which, out.ctypes.data, out.size)
        if rc == _lib.NONE:
            return None
        check(rc)
        return out

    @property
    def dtype(self):
        return np.uint64 if self.BITS == 64 else np.uint32

    def _arg(self, buf):
        ptr, count, esz, where, stream = buffer_info(buf)
        if esz != self.BITS // 8:
            raise TypeError("expected %d-byte elements" % (self.BITS // 8))
        return ptr, count, where, stream

    def _host(self, buf, what):
        """Argument of the reference's slice API: a host array (the C entry points take host pointers)."""
        ptr, count, where, _ = self._arg(buf)
        if where != _lib.MEM_HOST:
            raise TypeError("%s() takes host slices; use %s_batch() for device tensors" % (what, what))
        return ptr, count

    # -- the reference's slice API (host memory, one polynomial) -------------------------------
    def fwd(self, buf):
        ptr, count = self._host(buf, "fwd")
        check(getattr(lib(), self._p + "fwd")(self._h, ptr, count))

    def inv(self, buf):
        ptr, count = self._host(buf, "inv")
        check(getattr(lib(), self._p + "inv")(self._h, ptr, count))

    def mul_assign_normalize(self, lhs, rhs):
        lp, lc = self._host(lhs, "mul_assign_normalize")
        rp, rc_ = self._host(rhs, "mul_assign_normalize")
        check(getattr(lib(), self._p + "mul_assign_normalize")(self._h, lp, lc, rp, rc_))

    def normalize(self, values):
        vp, vc = self._host(values, "normalize")
        check(getattr(lib(), self._p + "normalize")(self._h, vp, vc))

    def mul_accumulate(self, acc, lhs, rhs):
        ap, ac = self._host(acc, "mul_accumulate")
        lp, lc = self._host(lhs, "mul_accumulate")
        rp, rc_ = self._host(rhs, "mul_accumulate")
        check(getattr(lib(), self._p + "mul_accumulate")(self._h, ap, ac, lp, lc, rp, rc_))

    # -- batched API: `batch` polynomials back to back, host arrays or device tensors ----------
    def _batch(self, buf):
        ptr, count, where, stream = self._arg(buf)
        if count % self._n:
            raise Panic("buffer length %d is not a multiple of ntt_size %d" % (count, self._n))
        return ptr, count // self._n, where, stream

    def fwd_batch(self, bufs):
        ptr, batch, where, stream = self._batch(bufs)
        check(getattr(lib(), self._p + "fwd_batch")(self._h, ptr, batch, where, stream))

    def inv_batch(self, bufs):
        ptr, batch, where, stream = self._batch(bufs)
        check(getattr(lib(), self._p + "inv_batch")(self._h, ptr, batch, where, stream))

    def mul_assign_normalize_batch(self, lhs, rhs):
        lp, batch, where, stream = self._batch(lhs)
        rp, rb, rwhere, _ = self._batch(rhs)
        if rb != batch or rwhere != where:
            raise Panic("lhs and rhs must have the same shape and live in the same memory")
        check(getattr(lib(), self._p + "mul_assign_normalize_batch")(self._h, lp, rp, batch, where, stream))

    def mul_ntt_batch(self, lhs, rhs_ntt):
        """Fused lhs <- inv(mul_assign_normalize(fwd(lhs), rhs_ntt)): same values as the three calls
        (src/prime64.rs:1254-1266), one pass over HBM for ntt_size <= 1024."""
        lp, batch, where, stream = self._batch(lhs)
        rp, rb, rwhere, _ = self._batch(rhs_ntt)
        if rb != batch or rwhere != where:
            raise Panic("lhs and rhs_ntt must have the same shape and live in the same memory")
        check(getattr(lib(), self._p + "mul_ntt_batch")(self._h, lp, rp, batch, where, stream))

    def external_product_batch(self, out, terms, key_ntt, nterms, nout, accumulate=False):
        """Fused mul_accumulate chain: out[b][o] (+)= inv(sum_j fwd(terms[b][j]) . key_ntt[j][o]) -- the values of
        fwd / mul_accumulate / inv (src/prime64.rs:794, :1085-1128, :872) called in sequence, in one pass over HBM."""
        op, ob, where, stream = self._batch(out)
        tp, tb, tw, _ = self._batch(terms)
        kp, kb, kw, _ = self._batch(key_ntt)
        if nout <= 0 or ob % nout or kb != nterms * nout or tb != (ob // nout) * nterms or tw != where or kw != where:
            raise Panic("out: batch*nout, terms: batch*nterms, key_ntt: nterms*nout polynomials in the same memory")
        check(getattr(lib(), self._p + "external_product_batch")(self._h, op, tp, kp, nterms, nout, ob // nout,
                                                                 1 if accumulate else 0, where, stream))

    def normalize_batch(self, values):
        vp, batch, where, stream = self._batch(values)
        check(getattr(lib(), self._p + "normalize_batch")(self._h, vp, batch, where, stream))

    def mul_accumulate_batch(self, acc, lhs, rhs):
        ap, batch, where, stream = self._batch(acc)
        lp, lb, lw, _ = self._batch(lhs)
        rp, rb, rw, _ = self._batch(rhs)
        if lb != batch or rb != batch or lw != where or rw != where:
            raise Panic("acc, lhs and rhs must have the same shape and live in the same memory")
        check(getattr(lib(), self._p + "mul_accumulate_batch")(self._h, ap, lp, rp, batch, where, stream))
