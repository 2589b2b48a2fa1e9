"""ctypes binding of libcntt_hip.so (include/cntt.h).  There is no fallback: if the HIP library is
missing or cannot be loaded this module raises, and every compute call needs a GPU."""
import ctypes
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libcntt_hip.so")

OK, NONE, EINVAL, ELEN, EDEVICE, ENOMEM = range(6)
MEM_HOST, MEM_DEVICE = 0, 1
TWID, TWID_SHOUP, INV_TWID, INV_TWID_SHOUP = range(4)

c_sz, c_u32, c_u64, c_vp, c_int, c_f = (ctypes.c_size_t, ctypes.c_uint32, ctypes.c_uint64, ctypes.c_void_p,
                                         ctypes.c_int, ctypes.c_float)


class PlanInfo(ctypes.Structure):
    _fields_ = [("ntt_size", c_u64), ("modulus", c_u64), ("p_barrett", c_u64), ("big_q", c_u64),
                ("n_inv_mod_p", c_u64), ("n_inv_mod_p_shoup", c_u64), ("root", c_u64),
                ("has_shoup", ctypes.c_int32), ("arith_class", ctypes.c_int32)]


class Panic(Exception):
    """The reference would panic here (assert_eq! on lengths, Div::new on modulus <= 1)."""


class DeviceError(RuntimeError):
    """HIP failure -- including 'no GPU present'.  There is no CPU path."""


def build(jobs=8):
    """Compile the HIP library in-tree (hipcc --offload-arch=gfx950)."""
    subprocess.run(["make", "-s", "-j%d" % jobs, "-C", os.path.join(_HERE, "csrc")], check=True)
    return LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError("libcntt_hip.so is not built: run `python -c 'import __graft_entry__ as g; g.build()'` "
                          "or `make -C concrete-ntt_amd/csrc` (there is no CPU fallback)")
    try:
        # PyTorch-ROCm bundles its own HIP runtime; load it first so that this library binds to the same
        # runtime instance (two HIP runtimes in one process leave torch without a device).
        import torch  # noqa: F401
    except ImportError:
        pass
    L = ctypes.CDLL(LIB_PATH)
    L.cntt_last_error.restype = ctypes.c_char_p
    L.cntt_version.restype = ctypes.c_char_p
    L.cntt_device_count.restype = c_int
    for bits, ct in ((64, c_u64), (32, c_u32)):
        p = "cntt_prime%d_" % bits
        sig = {
            "plan_new": (c_int, [c_sz, ct, c_vp]),
            "plan_clone": (c_vp, [c_vp]),
            "plan_free": (None, [c_vp]),
            "ntt_size": (c_sz, [c_vp]),
            "modulus": (ct, [c_vp]),
            "plan_info": (c_int, [c_vp, c_vp]),
            "plan_table": (c_int, [c_vp, c_int, c_vp, c_sz]),
            "fwd": (c_int, [c_vp, c_vp, c_sz]),
            "inv": (c_int, [c_vp, c_vp, c_sz]),
            "mul_assign_normalize": (c_int, [c_vp, c_vp, c_sz, c_vp, c_sz]),
            "normalize": (c_int, [c_vp, c_vp, c_sz]),
            "mul_accumulate": (c_int, [c_vp, c_vp, c_sz, c_vp, c_sz, c_vp, c_sz]),
            "fwd_batch": (c_int, [c_vp, c_vp, c_sz, c_int, c_vp]),
            "inv_batch": (c_int, [c_vp, c_vp, c_sz, c_int, c_vp]),
            "mul_assign_normalize_batch": (c_int, [c_vp, c_vp, c_vp, c_sz, c_int, c_vp]),
            "normalize_batch": (c_int, [c_vp, c_vp, c_sz, c_int, c_vp]),
            "mul_accumulate_batch": (c_int, [c_vp, c_vp, c_vp, c_vp, c_sz, c_int, c_vp]),
            "mul_ntt_batch": (c_int, [c_vp, c_vp, c_vp, c_sz, c_int, c_vp]),
            "external_product_batch": (c_int, [c_vp, c_vp, c_vp, c_vp, c_sz, c_sz, c_sz, c_int, c_int, c_vp]),
        }
        for name, (res, args) in sig.items():
            fn = getattr(L, p + name)
            fn.restype, fn.argtypes = res, args
    nsig = {
        "cntt_native_plan_new": (c_int, [c_int, c_sz, c_vp]),
        "cntt_native_plan_clone": (c_vp, [c_vp]),
        "cntt_native_plan_free": (None, [c_vp]),
        "cntt_native_ntt_size": (c_sz, [c_vp]),
        "cntt_native_nprimes": (c_int, [c_vp]),
        "cntt_native_word_bytes": (c_int, [c_vp]),
        "cntt_native_residue_bytes": (c_int, [c_vp]),
        "cntt_native_ntt32": (c_vp, [c_vp, c_int]),
        "cntt_native_ntt64": (c_vp, [c_vp, c_int]),
        "cntt_native_fwd": (c_int, [c_vp, c_vp, c_sz, c_vp]),
        "cntt_native_fwd_binary": (c_int, [c_vp, c_vp, c_sz, c_vp]),
        "cntt_native_inv": (c_int, [c_vp, c_vp, c_sz, c_vp]),
        "cntt_native_negacyclic_polymul": (c_int, [c_vp, c_vp, c_sz, c_vp, c_sz, c_vp, c_sz]),
        "cntt_native_fwd_batch": (c_int, [c_vp, c_vp, c_vp, c_sz, c_int, c_vp]),
        "cntt_native_fwd_binary_batch": (c_int, [c_vp, c_vp, c_vp, c_sz, c_int, c_vp]),
        "cntt_native_inv_batch": (c_int, [c_vp, c_vp, c_vp, c_sz, c_int, c_vp]),
        "cntt_native_negacyclic_polymul_batch": (c_int, [c_vp, c_vp, c_vp, c_vp, c_sz, c_int, c_vp]),
        "cntt_native_reserve": (c_int, [c_vp, c_sz]),
        "cntt_product_plan_new": (c_int, [c_sz, c_u64, c_vp, c_sz, c_vp]),
        "cntt_product_plan_clone": (c_vp, [c_vp]),
        "cntt_product_plan_free": (None, [c_vp]),
        "cntt_product_ntt_size": (c_sz, [c_vp]),
        "cntt_product_modulus": (c_u64, [c_vp]),
        "cntt_product_ntt_domain_len": (c_sz, [c_vp]),
        "cntt_product_nprimes32": (c_int, [c_vp]),
        "cntt_product_nprimes64": (c_int, [c_vp]),
        "cntt_product_prime": (c_u64, [c_vp, c_int]),
        "cntt_product_ntt32": (c_vp, [c_vp, c_int]),
        "cntt_product_ntt64": (c_vp, [c_vp, c_int]),
        "cntt_product_modular_inverses": (c_int, [c_vp, c_vp, c_sz]),
        "cntt_product_fwd": (c_int, [c_vp, c_vp, c_sz, c_vp, c_sz, c_int, c_u64]),
        "cntt_product_inv": (c_int, [c_vp, c_vp, c_sz, c_vp, c_sz, c_int]),
        "cntt_product_mul_assign_normalize": (c_int, [c_vp, c_vp, c_sz, c_vp, c_sz]),
        "cntt_product_normalize": (c_int, [c_vp, c_vp, c_sz]),
        "cntt_product_mul_accumulate": (c_int, [c_vp, c_vp, c_sz, c_vp, c_sz, c_vp, c_sz]),
        "cntt_product_fwd_batch": (c_int, [c_vp, c_vp, c_vp, c_sz, c_int, c_u64, c_int, c_vp]),
        "cntt_product_inv_batch": (c_int, [c_vp, c_vp, c_vp, c_sz, c_int, c_int, c_vp]),
        "cntt_product_mul_assign_normalize_batch": (c_int, [c_vp, c_vp, c_vp, c_sz, c_int, c_vp]),
        "cntt_product_normalize_batch": (c_int, [c_vp, c_vp, c_sz, c_int, c_vp]),
        "cntt_product_mul_accumulate_batch": (c_int, [c_vp, c_vp, c_vp, c_vp, c_sz, c_int, c_vp]),
        "cntt_product_external_product_batch": (c_int, [c_vp, c_vp, c_vp, c_vp, c_sz, c_sz, c_sz, c_int, c_u64, c_int,
                                                        c_int, c_vp]),
        "cntt_debug_set": (c_int, [ctypes.c_char_p, c_int]),
        "cntt_debug_get": (c_int, [ctypes.c_char_p, c_vp]),
        "cntt_shard_bounds": (c_int, [c_sz, c_int, c_int, c_vp, c_vp]),
        "cntt_fill_uniform_u64": (c_int, [c_vp, c_sz, c_u64, c_u64, c_vp]),
        "cntt_fill_uniform_u32": (c_int, [c_vp, c_sz, c_u32, c_u64, c_vp]),
    }
    for name, (res, args) in nsig.items():
        fn = getattr(L, name)
        fn.restype, fn.argtypes = res, args
    _lib = L
    return L


def last_error():
    return lib().cntt_last_error().decode()


def check(rc):
    """Status code -> the reference's behaviour: None stays a value, panics and device errors raise."""
    if rc == OK:
        return
    msg = last_error()
    if rc in (EINVAL, ELEN):
        raise Panic(msg)
    if rc == ENOMEM:
        raise MemoryError(msg)
    if rc == EDEVICE:
        raise DeviceError(msg)
    raise RuntimeError("unexpected status %d: %s" % (rc, msg))


def buffer_info(buf):
    """(pointer, element count, element size, where, stream) of a numpy array or a torch tensor."""
    if hasattr(buf, "data_ptr"):  # torch tensor
        if not buf.is_contiguous():
            raise ValueError("tensor must be contiguous")
        where = MEM_DEVICE if buf.is_cuda else MEM_HOST
        stream = None
        if buf.is_cuda:
            import torch
            # the library keys its per-device plan replicas on the CURRENT HIP device and launches there
            if buf.device.index != torch.cuda.current_device():
                raise ValueError("tensor lives on cuda:%d but the current device is cuda:%d: call "
                                 "torch.cuda.set_device (or use torch.cuda.device) first"
                                 % (buf.device.index, torch.cuda.current_device()))
            stream = torch.cuda.current_stream(buf.device).cuda_stream
        return buf.data_ptr(), buf.numel(), buf.element_size(), where, stream
    if not buf.flags["C_CONTIGUOUS"]:
        raise ValueError("array must be C-contiguous")
    return buf.ctypes.data, buf.size, buf.itemsize, MEM_HOST, None
