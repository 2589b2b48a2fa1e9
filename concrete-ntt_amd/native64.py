"""concrete_ntt::native64 (src/native64.rs)."""
from ._native import _make

Plan32 = _make(1, 5, 8, 4, False, "native64::Plan32 (src/native64.rs): 5 x 30-bit primes")
# The reference offers Plan52 only with AVX-512 IFMA (nightly); here it runs on the u64 HIP kernels.
Plan52 = _make(7, 3, 8, 8, False, "native64::Plan52 (src/native64.rs): 3 x 50-bit primes")
