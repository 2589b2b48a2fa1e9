"""native32/64/128 and native_binary32/64/128 plans over the C ABI (shared implementation)."""
import ctypes

import numpy as np

from . import _lib
from ._lib import Panic, buffer_info, check, lib
from .prime32 import Plan as _Plan32
from .prime64 import Plan as _Plan64


class NativePlan:
    """Mirrors the reference's PlanNN types: try_new(n), ntt_size(), ntt_i(), fwd, fwd_binary, inv,
    negacyclic_polymul (e.g. src/native64.rs:930-1070).  Coefficients: numpy uint32 / uint64 arrays;
    u128 words are (lo, hi) uint64 pairs, i.e. arrays of 2n uint64 (16-byte little-endian)."""

    KIND = 1
    NPRIMES, WORD, RES = 5, 8, 4
    BINARY = False

    def __init__(self, handle):
        self._h = handle
        self._n = lib().cntt_native_ntt_size(handle)

    @classmethod
    def try_new(cls, n):
        out = ctypes.c_void_p()
        rc = lib().cntt_native_plan_new(cls.KIND, n, ctypes.byref(out))
        if rc == _lib.NONE:
            return None
        check(rc)
        return cls(out.value)

    def clone(self):
        return type(self)(lib().cntt_native_plan_clone(self._h))

    def __del__(self):
        try:
            if self._h:
                lib().cntt_native_plan_free(self._h)
        except Exception:
            pass

    def ntt_size(self):
        return self._n

    def ntt(self, i):
        """ntt_0() .. ntt_k(): borrowed prime sub-plan (src/native64.rs:950-969)."""
        if self.RES == 8:
            h = lib().cntt_native_ntt64(self._h, i)
            return _Plan64(h, owned=False, parent=self) if h else None
        h = lib().cntt_native_ntt32(self._h, i)
        return _Plan32(h, owned=False, parent=self) if h else None

    # -- helpers -----------------------------------------------------------------------------
    @property
    def word_dtype(self):
        return np.uint32 if self.WORD == 4 else np.uint64

    @property
    def res_dtype(self):
        return np.uint64 if self.RES == 8 else np.uint32

    def _words(self, buf):
        ptr, count, esz, where, stream = buffer_info(buf)
        if esz != min(self.WORD, 8):
            raise TypeError("expected %d-byte words" % self.WORD)
        return ptr, count // (2 if self.WORD == 16 else 1), where, stream

    def _res(self, residues, where, count):
        if len(residues) != self.NPRIMES:
            raise Panic("expected %d residue buffers" % self.NPRIMES)
        ptrs = []
        for r in residues:
            ptr, c, esz, w, _ = buffer_info(r)
            if esz != self.RES or w != where or c != count:
                raise Panic("residue buffers must match the value buffer (size, memory)")
            ptrs.append(ptr)
        return (ctypes.c_void_p * self.NPRIMES)(*ptrs)

    # -- the reference's slice API (host memory: the C entry points take host pointers) -------------
    def _host_words(self, buf, what):
        ptr, count, where, _ = self._words(buf)
        if where != _lib.MEM_HOST:
            raise TypeError("%s() takes host slices; use %s_batch() for device tensors" % (what, what))
        return ptr, count, where

    def fwd(self, value, *residues):
        ptr, count, where = self._host_words(value, "fwd")
        check(lib().cntt_native_fwd(self._h, ptr, count, self._res(residues, where, count)))

    def fwd_binary(self, value, *residues):
        if not self.BINARY:
            raise AttributeError("fwd_binary exists only on native_binary* plans")
        ptr, count, where = self._host_words(value, "fwd")
        check(lib().cntt_native_fwd_binary(self._h, ptr, count, self._res(residues, where, count)))

    def inv(self, value, *residues):
        ptr, count, where = self._host_words(value, "inv")
        check(lib().cntt_native_inv(self._h, ptr, count, self._res(residues, where, count)))

    def negacyclic_polymul(self, prod, lhs, rhs):
        pp, pc, _ = self._host_words(prod, "negacyclic_polymul")
        lp, lc, _ = self._host_words(lhs, "negacyclic_polymul")
        rp, rc_, _ = self._host_words(rhs, "negacyclic_polymul")
        check(lib().cntt_native_negacyclic_polymul(self._h, pp, pc, lp, lc, rp, rc_))

    # -- batched --------------------------------------------------------------------------------
    def _batchn(self, buf):
        ptr, count, where, stream = self._words(buf)
        if count % self._n:
            raise Panic("buffer length is not a multiple of ntt_size")
        return ptr, count, count // self._n, where, stream

    def fwd_batch(self, value, residues, binary=False):
        ptr, count, batch, where, stream = self._batchn(value)
        fn = lib().cntt_native_fwd_binary_batch if binary else lib().cntt_native_fwd_batch
        check(fn(self._h, ptr, self._res(residues, where, count), batch, where, stream))

    def inv_batch(self, value, residues):
        ptr, count, batch, where, stream = self._batchn(value)
        check(lib().cntt_native_inv_batch(self._h, ptr, self._res(residues, where, count), batch, where, stream))

    def negacyclic_polymul_batch(self, prod, lhs, rhs):
        pp, pc, batch, where, stream = self._batchn(prod)
        lp, lc, _, lw, _ = self._batchn(lhs)
        rp, rc_, _, rw, _ = self._batchn(rhs)
        if lc != pc or rc_ != pc or lw != where or rw != where:
            raise Panic("prod, lhs and rhs must have the same shape and live in the same memory")
        check(lib().cntt_native_negacyclic_polymul_batch(self._h, pp, lp, rp, batch, where, stream))

    def reserve(self, batch):
        check(lib().cntt_native_reserve(self._h, batch))


def _make(kind, nprimes, word, res, binary, doc):
    return type("Plan", (NativePlan,), {"KIND": kind, "NPRIMES": nprimes, "WORD": word, "RES": res,
                                        "BINARY": binary, "__doc__": doc})
