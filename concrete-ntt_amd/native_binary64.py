"""concrete_ntt::native_binary64 (src/native_binary64.rs)."""
from ._native import _make

Plan32 = _make(4, 3, 8, 4, True, "native_binary64::Plan32 (src/native_binary64.rs): 3 x 30-bit primes")
# The reference offers Plan52 only with AVX-512 IFMA (nightly); here it runs on the u64 HIP kernels.
Plan52 = _make(9, 2, 8, 8, True, "native_binary64::Plan52 (src/native_binary64.rs): 2 x 50-bit primes")
