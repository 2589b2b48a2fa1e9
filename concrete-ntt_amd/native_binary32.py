"""concrete_ntt::native_binary32 (src/native_binary32.rs)."""
from ._native import _make

Plan32 = _make(3, 2, 4, 4, True, "native_binary32::Plan32 (src/native_binary32.rs): 2 x 30-bit primes")
# The reference offers Plan52 only with AVX-512 IFMA (nightly); here it runs on the u64 HIP kernels.
Plan52 = _make(8, 1, 4, 8, True, "native_binary32::Plan52 (src/native_binary32.rs): 1 x 50-bit primes")
