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
