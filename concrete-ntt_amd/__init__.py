"""MI355X-native batched negacyclic NTT engine with the API surface of zama-ai/concrete-ntt.

Module layout mirrors the reference crate (src/lib.rs:88-110): prime32, prime64, native32, native64,
native128, native_binary32, native_binary64, native_binary128, product.  Every transform runs in hand-written
HIP kernels (csrc/) behind the C ABI of include/cntt.h; there is no CPU compute path.
"""
from . import _lib
from ._lib import DeviceError, Panic, build, lib  # noqa: F401
from . import prime32, prime64  # noqa: F401
from . import native32, native64, native128, native_binary32, native_binary64, native_binary128  # noqa: F401
from . import product  # noqa: F401


def device_count():
    return lib().cntt_device_count()


def version():
    return lib().cntt_version().decode()


def build_info():
    """{'version': ..., 'csrc_hash': ...}: the hash of the csrc/ sources the loaded library was built from."""
    v = version()
    return {"version": v, "csrc_hash": v.split("csrc:")[1].strip() if "csrc:" in v else None}


def fill_uniform(tensor, bound, seed):
    """Synthetic input generated on the device into a 4- or 8-byte-element tensor."""
    ptr, count, esz, where, stream = _lib.buffer_info(tensor)
    if where != _lib.MEM_DEVICE:
        raise TypeError("fill_uniform needs a device tensor")
    fn = lib().cntt_fill_uniform_u64 if esz == 8 else lib().cntt_fill_uniform_u32
    _lib.check(fn(ptr, count, bound, seed, stream))


def debug_set(key, value):
    """TESTING ONLY (include/cntt.h): kernel-selection switch `key` := value (-1: the library's default; key "reset": all defaults).
    Results are identical for every setting.  The two class switches ("fp", "pm64") are read when a plan is created."""
    _lib.check(lib().cntt_debug_set(key.encode(), int(value)))


def debug_get(key):
    import ctypes
    v = ctypes.c_int(0)
    _lib.check(lib().cntt_debug_get(key.encode(), ctypes.addressof(v)))
    return v.value


class debug_switches:
    """with cntt.debug_switches(fp=0): ...  -- sets the switches for the block and restores the previous values afterwards."""

    def __init__(self, **kw):
        self._kw, self._old = kw, {}

    def __enter__(self):
        for k, v in self._kw.items():
            self._old[k] = debug_get(k)
            debug_set(k, v)
        return self

    def __exit__(self, *exc):
        for k, v in self._old.items():
            debug_set(k, v)
        return False


def shard_bounds(batch, world, rank):
    """[begin, end) of `rank`'s contiguous shard through the C ABI (cntt_shard_bounds; shard.py holds the same arithmetic in Python)."""
    import ctypes
    b, e = ctypes.c_size_t(0), ctypes.c_size_t(0)
    _lib.check(lib().cntt_shard_bounds(batch, world, rank, ctypes.addressof(b), ctypes.addressof(e)))
    return b.value, e.value
